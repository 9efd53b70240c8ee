// iupac.hip -- K4: IUPAC degenerate-pattern aggregation (gfx950).
//
// Replaces IUPACPattern::basepatterns_from_iupac_{single,double}_stranded + the summation loop of
// aggregate_attributes_from_basepatterns / count_combined_occurences
// (src/iupac_pattern.cpp:331-473,806-833).
//
//  * PLUS: the reference's LIFO stack emits an odometer in which the LAST degenerate position moves
//    fastest and every position runs rep[0], rep[n-1], ..., rep[1]; count / background / expected are
//    added in exactly that order (the float32 sums are order-sensitive).
//  * BOTH: the reference canonicalises (min(id, rc)), sorts ascending and skips adjacent
//    duplicates, i.e. it sums over the DISTINCT canonical ids in ascending order.
//
// Only the float32 fold is inherently serial.  Everything else is parallel: a workgroup per small
// pattern (ids sorted in LDS), a grid-wide bitmap / compaction / gather pipeline per large pattern, and
// a v_readlane chain that folds 64 gathered values at a time in the reference's order.
// (Round 1 started with one thread per pattern: 3.2 s of hill-climb on MafK W=10; now 0.1 s.)
//
// The three sums come back to the host; z-score and log-p (O(1) scalar double math per pattern,
// :446-470) are finished in launch_iupac with the same libm the reference links.
#include <math.h>

#include <algorithm>
#include <limits>
#include <new>
#include <vector>

#include "pengk_internal.h"

namespace pengk {
namespace {

__constant__ int c_rep_n[11] = {1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 4};
__constant__ int c_rep[11][4] = {{0, 0, 0, 0}, {1, 1, 1, 1}, {2, 2, 2, 2}, {3, 3, 3, 3}, {1, 2, 0, 0}, {0, 3, 0, 0},
                                 {0, 2, 0, 0}, {1, 3, 0, 0}, {0, 1, 0, 0}, {2, 3, 0, 0}, {0, 1, 2, 3}};
__constant__ int c_comp[11] = {3, 2, 1, 0, 4, 5, 7, 6, 9, 8, 10};  // A<->T C<->G S W R<->Y M<->K N

struct RawSums {
  unsigned long long sites;
  float bg_p;
  float expected;
};

constexpr int MAXW = PENGK_MAX_W;

// ---------------------------------------------------------------------------------------------
// K4 v2: one 256-thread workgroup per pattern.  The float32 sums must be formed in the reference's
// order, but nothing else has to be serial: all threads generate / sort the ids and gather the three
// table values; then one wave folds 64 values at a time with a v_readlane chain (2 dependent adds per
// element instead of three dependent HBM/L2 gathers per element in the thread-per-pattern kernel).
//   PLUS: element k of the LIFO odometer is unranked directly (letter sets have 1, 2 or 4 members,
//         so the mixed radix is bit slicing).
//   BOTH: min(id, rc(id)) of all members is sorted in LDS (bitonic), adjacent duplicates contribute 0.
// Patterns with more than IUPAC_LDS_MAX members fall back to the thread-per-pattern kernel.
// ---------------------------------------------------------------------------------------------
constexpr int IUPAC_LDS_MAX = 8192;

__device__ __forceinline__ float chain_add(float acc, float v) {
  // acc + v[0] + v[1] + ... + v[63], strictly left to right (wave-uniform result)
#pragma unroll
  for (int i = 0; i < 64; ++i) acc += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), i));
  return acc;
}

__global__ __launch_bounds__(256) void iupac_block_kernel(int W, int both, const unsigned long long* __restrict__ ids, int n_pat,
                                                          const uint32_t* __restrict__ counts,
                                                          const float* __restrict__ bgp, const float* __restrict__ expected,
                                                          RawSums* __restrict__ out) {
  __shared__ uint32_t s_ids[IUPAC_LDS_MAX];
  __shared__ float s_bg[256], s_ex[256];
  __shared__ unsigned long long s_cnt;
  __shared__ int s_letters[MAXW];
  const int pat = blockIdx.x;
  const int tid = threadIdx.x;
  if (tid == 0) {
    unsigned long long t = ids[pat];
    for (int p = 0; p < W; ++p) {
      s_letters[p] = (int)(t % 11ull);
      t /= 11ull;
    }
    s_cnt = 0;
  }
  __syncthreads();
  int lg[MAXW];  // log2 of the letter-set sizes
  int lgn = 0;
  for (int p = 0; p < W; ++p) {
    const int sz = c_rep_n[s_letters[p]];
    lg[p] = sz == 1 ? 0 : (sz == 2 ? 1 : 2);
    lgn += lg[p];
  }
  const uint32_t n = 1u << lgn;
  if (n > (uint32_t)IUPAC_LDS_MAX) return;  // block-uniform: launch_iupac runs the list pipeline for this one
  // member k in the reference's emission order (last degenerate position fastest; rep[0], rep[n-1], ..., rep[1])
  auto member = [&](uint32_t k) -> uint32_t {
    uint32_t x = 0;
    for (int p = W - 1; p >= 0; --p) {
      const int sz = 1 << lg[p];
      const int d = (int)(k & (uint32_t)(sz - 1));
      k >>= lg[p];
      const int j = d == 0 ? 0 : sz - d;
      x |= (uint32_t)c_rep[s_letters[p]][j] << (2 * p);
    }
    return x;
  };
  uint32_t npad = n;
  if (both) {
    for (uint32_t k = tid; k < n; k += 256) {
      const uint32_t x = member(k);
      const uint32_t r = revcomp32(x, W);
      s_ids[k] = x < r ? x : r;
    }
    __syncthreads();
    // bitonic sort, ascending (n is a power of two)
    for (uint32_t size = 2; size <= n; size <<= 1)
      for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
        for (uint32_t i = tid; i < (n >> 1); i += 256) {
          const uint32_t lo = 2 * i - (i & (stride - 1));  // index with bit `stride` clear
          const uint32_t hi = lo + stride;
          const bool up = (lo & size) == 0;
          const uint32_t a = s_ids[lo], b = s_ids[hi];
          if ((a > b) == up) {
            s_ids[lo] = b;
            s_ids[hi] = a;
          }
        }
        __syncthreads();
      }
  }
  (void)npad;
  float sum_bg = 0.0f, sum_e = 0.0f;
  unsigned long long my_cnt = 0;
  for (uint32_t base = 0; base < n; base += 256) {
    const uint32_t k = base + tid;
    float vb = 0.0f, ve = 0.0f;
    if (k < n) {
      uint32_t x;
      bool take = true;
      if (both) {
        x = s_ids[k];
        take = (k == 0) || (s_ids[k - 1] != x);
      } else {
        x = member(k);
      }
      if (take) {
        vb = bgp[x];
        ve = expected[x];
        my_cnt += counts[x];
      }
    }
    s_bg[tid] = vb;
    s_ex[tid] = ve;
    __syncthreads();
    if (tid < 64) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (base + 64u * q < n) {  // wave-uniform
          sum_bg = chain_add(sum_bg, s_bg[q * 64 + tid]);
          sum_e = chain_add(sum_e, s_ex[q * 64 + tid]);
        }
      }
    }
    __syncthreads();
  }
  if (my_cnt) atomicAdd(&s_cnt, my_cnt);
  __syncthreads();
  if (tid == 0) {
    out[pat].sites = s_cnt;
    out[pat].bg_p = sum_bg;
    out[pat].expected = sum_e;
  }
}


// ---------------------------------------------------------------------------------------------
// K4 for patterns too large for one workgroup's LDS (thousands to 4^W members; N-rich mutants of the
// hill-climb).  Everything except the float32 fold is data-parallel:
//   BOTH: mark min(id, rc) of every member in a 4^W-bit map (the map IS the sorted, de-duplicated
//         list), compact it to an ascending id array with a popcount scan;
//   PLUS: no list -- member k of the emission order is unranked on the fly;
//   then gather the table values in list order (grid-wide) and fold them 64 at a time in one wave.
// ---------------------------------------------------------------------------------------------
struct PatternSpec {
  unsigned long long letters4;  // 4 bits per position
  int W;
  int lgn;
};

__device__ __forceinline__ uint32_t spec_member(const PatternSpec& ps, uint32_t k) {
  uint32_t x = 0;
  for (int p = ps.W - 1; p >= 0; --p) {
    const int L = (int)((ps.letters4 >> (4 * p)) & 15ull);
    const int sz = c_rep_n[L];
    const int lg = sz == 1 ? 0 : (sz == 2 ? 1 : 2);
    const int d = (int)(k & (uint32_t)(sz - 1));
    k >>= lg;
    x |= (uint32_t)c_rep[L][d == 0 ? 0 : sz - d] << (2 * p);
  }
  return x;
}

// One "slot" per large pattern of a group; blockIdx.y selects the slot, so all large patterns of a group run side
// by side (the fold is one wave per pattern: 50 N-rich mutants of a hill-climb round fold on 50 CUs at once
// instead of one after the other).  Slot y owns bitmap[y * n_words ..), lists [y * cap ..), head[y].
struct BigSlot {
  PatternSpec ps;
  uint32_t out_index;  // row of the output this pattern belongs to
  uint32_t pad;
};
struct BigHead {  // zeroed before every group
  uint32_t m;     // BOTH: number of distinct canonical ids
  uint32_t pad;
  unsigned long long cnt;
};

__global__ __launch_bounds__(256) void iupac_mark_kernel(const BigSlot* __restrict__ slots, uint32_t* __restrict__ bitmaps,
                                                         uint32_t n_words) {
  const PatternSpec ps = slots[blockIdx.y].ps;
  uint32_t* bitmap = bitmaps + (size_t)blockIdx.y * n_words;
  const uint32_t n = 1u << ps.lgn;
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    const uint32_t x = spec_member(ps, k);
    const uint32_t r = revcomp32(x, ps.W);
    const uint32_t c = x < r ? x : r;
    atomicOr(&bitmap[c >> 5], 1u << (c & 31u));
  }
}

// one workgroup per slot: ascending ids of the set bits -> the slot's id list, their number -> head.m
__global__ __launch_bounds__(1024) void iupac_compact_kernel(const uint32_t* __restrict__ bitmaps, uint32_t n_words,
                                                             uint32_t* __restrict__ lists, size_t cap, BigHead* __restrict__ heads) {
  __shared__ uint32_t s_scan[1024];
  const uint32_t* bitmap = bitmaps + (size_t)blockIdx.y * n_words;
  uint32_t* out_ids = lists + (size_t)blockIdx.y * cap;
  const uint32_t per = (n_words + 1023u) / 1024u;
  const uint32_t w0 = min(n_words, threadIdx.x * per), w1 = min(n_words, w0 + per);
  uint32_t mine = 0;
  for (uint32_t w = w0; w < w1; ++w) mine += __popc(bitmap[w]);
  s_scan[threadIdx.x] = mine;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
    const uint32_t v = threadIdx.x >= off ? s_scan[threadIdx.x - off] : 0u;
    __syncthreads();
    s_scan[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t at = s_scan[threadIdx.x] - mine;
  for (uint32_t w = w0; w < w1; ++w) {
    uint32_t bits = bitmap[w];
    while (bits) {
      const uint32_t b = __ffs(bits) - 1;
      bits &= bits - 1;
      out_ids[at++] = (w << 5) | b;
    }
  }
  if (threadIdx.x == 1023) heads[blockIdx.y].m = s_scan[1023];
}

// values in list order; use_list == 0: list = emission order of the pattern (PLUS), m = 2^lgn
__global__ __launch_bounds__(256) void iupac_gather_kernel(const BigSlot* __restrict__ slots, int use_list,
                                                           const uint32_t* __restrict__ lists, size_t cap,
                                                           const uint32_t* __restrict__ counts, const float* __restrict__ bgp,
                                                           const float* __restrict__ expected, float* __restrict__ vbs,
                                                           float* __restrict__ ves, BigHead* __restrict__ heads) {
  const PatternSpec ps = slots[blockIdx.y].ps;
  const uint32_t* ids = lists + (size_t)blockIdx.y * cap;
  float* vb = vbs + (size_t)blockIdx.y * cap;
  float* ve = ves + (size_t)blockIdx.y * cap;
  const uint32_t m = use_list ? heads[blockIdx.y].m : (1u << ps.lgn);
  unsigned long long mine = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
    const uint32_t x = use_list ? ids[i] : spec_member(ps, i);
    vb[i] = bgp[x];
    ve[i] = expected[x];
    mine += counts[x];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&heads[blockIdx.y].cnt, mine);
}

// one wave per slot: strictly sequential float32 sums of the slot's vb[0..m) and ve[0..m)
__global__ __launch_bounds__(64) void iupac_fold_kernel(const BigSlot* __restrict__ slots, int use_list, const float* __restrict__ vbs,
                                                        const float* __restrict__ ves, size_t cap,
                                                        const BigHead* __restrict__ heads, RawSums* __restrict__ out) {
  const BigSlot slot = slots[blockIdx.x];
  const float* vb = vbs + (size_t)blockIdx.x * cap;
  const float* ve = ves + (size_t)blockIdx.x * cap;
  const uint32_t m = use_list ? heads[blockIdx.x].m : (1u << slot.ps.lgn);
  const uint32_t lane = threadIdx.x;
  float sb = 0.0f, se = 0.0f;
  float nb = lane < m ? vb[lane] : 0.0f, ne = lane < m ? ve[lane] : 0.0f;
  for (uint32_t base = 0; base < m; base += 64) {
    const float cb = nb, ce = ne;
    const uint32_t nx = base + 64 + lane;  // prefetch the next 64 while this chain runs
    nb = nx < m ? vb[nx] : 0.0f;
    ne = nx < m ? ve[nx] : 0.0f;
    sb = chain_add(sb, cb);
    se = chain_add(se, ce);
  }
  if (lane == 0) {
    RawSums* o = out + slot.out_index;
    o->sites = heads[blockIdx.x].cnt;
    o->bg_p = sb;
    o->expected = se;
  }
}

float log_bonferroni(int letter) {  // src/iupac_pattern.cpp:199-210
  if (letter < 4) return (float)log(8.0);
  if (letter < 8) return (float)log(16.0);
  if (letter < 10) return (float)log(24.0);
  return (float)log(6.0);
}

}  // namespace

int launch_iupac(pengk_ctx* ctx, int W, int both, const uint64_t* h_ids, int64_t n, const uint32_t* d_counts,
                 const float* d_bgp, const float* d_expected, pengk_iupac_stats* h_out) {
  const size_t id_bytes = (size_t)n * sizeof(uint64_t);
  const size_t out_bytes = (size_t)n * sizeof(RawSums);
  const size_t id_pad = (id_bytes + 255) & ~(size_t)255;
  const size_t out_pad = (out_bytes + 255) & ~(size_t)255;
  int rc = ensure_scratch(ctx, &ctx->d_misc, &ctx->misc_bytes, id_pad + out_pad);
  if (rc) return rc;
  unsigned long long* d_ids = (unsigned long long*)ctx->d_misc;
  RawSums* d_out = (RawSums*)((char*)ctx->d_misc + id_pad);
  PENGK_HIP(hipMemcpyAsync(d_ids, h_ids, id_bytes, hipMemcpyHostToDevice, ctx->stream));
  // small patterns (<= IUPAC_LDS_MAX members): one workgroup each, all in one launch
  hipLaunchKernelGGL(iupac_block_kernel, dim3((unsigned)n), dim3(256), 0, ctx->stream, W, both, d_ids, (int)n, d_counts, d_bgp,
                     d_expected, d_out);
  PENGK_HIP(hipGetLastError());
  // large patterns: the data-parallel list pipeline, a group of patterns per round of launches
  static const int rep_n[11] = {1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 4};
  std::vector<BigSlot> big;
  for (int64_t i = 0; i < n; ++i) {
    PatternSpec ps{0ull, W, 0};
    uint64_t t = h_ids[i];
    for (int p = 0; p < W; ++p) {
      const int L = (int)(t % 11);
      t /= 11;
      ps.letters4 |= (unsigned long long)L << (4 * p);
      ps.lgn += rep_n[L] == 1 ? 0 : (rep_n[L] == 2 ? 1 : 2);
    }
    if ((1ull << ps.lgn) > (unsigned long long)IUPAC_LDS_MAX) big.push_back(BigSlot{ps, (uint32_t)i, 0u});
  }
  const uint32_t n_words = 1u << (2 * W - 5);
  const size_t bm_bytes = both ? (size_t)n_words * 4 : 0;
  const size_t budget = ctx->iupac_group_bytes ? (size_t)ctx->iupac_group_bytes : (size_t)1 << 30;  // scratch for one group
  for (size_t g0 = 0; g0 < big.size();) {
    // longest run of patterns whose slots (all sized for the largest member list among them) fit the budget
    size_t g1 = g0, cap = 0;
    while (g1 < big.size()) {
      const size_t c = std::max(cap, (size_t)1 << big[g1].ps.lgn);
      const size_t per_slot = bm_bytes + sizeof(BigSlot) + sizeof(BigHead) + 3 * c * 4;
      if (g1 > g0 && ((g1 - g0 + 1) * per_slot > budget || g1 - g0 >= 65535)) break;
      cap = c;
      ++g1;
    }
    const size_t G = g1 - g0;
    // scratch: slots | heads | bitmaps | id lists | vb | ve   (every part 256-byte aligned)
    auto pad256 = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_heads = pad256(G * sizeof(BigSlot));
    const size_t o_bm = o_heads + pad256(G * sizeof(BigHead));
    const size_t o_lst = o_bm + pad256(G * bm_bytes);
    const size_t o_vb = o_lst + pad256(G * cap * 4);
    const size_t o_ve = o_vb + pad256(G * cap * 4);
    rc = ensure_scratch(ctx, &ctx->d_iupac_big, &ctx->iupac_big_bytes, o_ve + pad256(G * cap * 4));
    if (rc) return rc;
    char* base = (char*)ctx->d_iupac_big;
    BigSlot* d_slots = (BigSlot*)base;
    BigHead* d_heads = (BigHead*)(base + o_heads);
    uint32_t* bitmaps = (uint32_t*)(base + o_bm);
    uint32_t* lists = (uint32_t*)(base + o_lst);
    float* vbs = (float*)(base + o_vb);
    float* ves = (float*)(base + o_ve);
    // the slot table is read by kernels of THIS group only after the copy; the previous group's kernels are ahead of it in
    // the stream, and `big` outlives the synchronisation at the end of this function
    PENGK_HIP(hipMemcpyAsync(d_slots, big.data() + g0, G * sizeof(BigSlot), hipMemcpyHostToDevice, ctx->stream));
    PENGK_HIP(hipMemsetAsync(d_heads, 0, G * sizeof(BigHead), ctx->stream));
    const unsigned gblocks = (unsigned)std::min<size_t>(std::max<size_t>(cap / 256, 1), 2048);
    if (both) {
      PENGK_HIP(hipMemsetAsync(bitmaps, 0, G * bm_bytes, ctx->stream));
      hipLaunchKernelGGL(iupac_mark_kernel, dim3(gblocks, (unsigned)G), dim3(256), 0, ctx->stream, d_slots, bitmaps, n_words);
      hipLaunchKernelGGL(iupac_compact_kernel, dim3(1, (unsigned)G), dim3(1024), 0, ctx->stream, bitmaps, n_words, lists, cap, d_heads);
    }
    hipLaunchKernelGGL(iupac_gather_kernel, dim3(gblocks, (unsigned)G), dim3(256), 0, ctx->stream, d_slots, both, lists, cap, d_counts,
                       d_bgp, d_expected, vbs, ves, d_heads);
    hipLaunchKernelGGL(iupac_fold_kernel, dim3((unsigned)G), dim3(64), 0, ctx->stream, d_slots, both, vbs, ves, cap, d_heads, d_out);
    PENGK_HIP(hipGetLastError());
    g0 = g1;
  }
  RawSums* raw = new (std::nothrow) RawSums[(size_t)n];
  if (!raw) return fail(PENGK_ERR_NOMEM, "pengk_iupac_aggregate: out of host memory");
  hipError_t e = hipMemcpyAsync(raw, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    delete[] raw;
    return hip_fail(e, "pengk_iupac_aggregate");
  }
  for (int64_t i = 0; i < n; ++i) {
    pengk_iupac_stats& o = h_out[i];
    const unsigned long long sum_c = raw[i].sites;
    const float sum_e = raw[i].expected;
    o.sites = sum_c;
    o.bg_p = raw[i].bg_p;
    o.expected = sum_e;
    o.zscore = (float)((double)((float)sum_c - sum_e) / sqrt((double)sum_e));  // :446
    if (sum_c == 0) {
      o.log_pvalue = std::numeric_limits<float>::infinity();
    } else {
      const float mu = sum_e;
      const float frac = 1 - mu / (float)(sum_c + 1);  // float arithmetic (:457)
      float lp = 0;
      if ((float)sum_c > mu && sum_c > 5 && o.zscore > 2)
        lp = (float)((double)sum_c * log((double)(mu / (float)sum_c)) + (double)sum_c - (double)mu -
                     0.5 * log(6.283 * (double)sum_c * (double)frac * (double)frac));
      uint64_t t = h_ids[i];
      for (int p = 0; p < W; ++p) {
        lp += log_bonferroni((int)(t % 11));
        t /= 11;
      }
      o.log_pvalue = lp;
    }
  }
  delete[] raw;
  return PENGK_OK;
}

// (pengk_warmup: loads this translation unit's code object ahead of its first launch)
int warm_iupac() {
  hipFuncAttributes a;
  PENGK_HIP(hipFuncGetAttributes(&a, (const void*)iupac_block_kernel));
  return PENGK_OK;
}

}  // namespace pengk
