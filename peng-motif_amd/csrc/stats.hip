// stats.hip -- K2+K3: the pattern-space sweep, one thread per pattern (gfx950).
//
// Replaces BasePattern::calculate_bg_probabilities / calculate_bg_probability (recursive, OpenMP
// over the 4^(k+1) initial mers, src/base_pattern.cpp:285-325), aggregate_double_strand_background
// (:268-283), calculate_expected_counts (:260-265), calculate_log_pvalues (:231-250) and
// calculate_zscores (:252-258) by ONE pass: each thread derives the order-0..max_k probabilities of
// its pattern AND of the reverse complement from the 84-entry V table in LDS (float32 products in
// position order, so every value has the reference's bits), adds the twins, and writes
// bgprob[0..max_k], expected, log-p and z.  28 B per pattern at max_k = 2: HBM-write bound.
//
// Float semantics follow SURVEY.md A.4: mu/n and the products are float32; sqrt, log and the
// z division are double (the reference's unqualified calls resolve to the double overloads).
// Compiled with -ffp-contract=off so no multiply-add is fused.
#include "pengk_internal.h"

#include <vector>

namespace pengk {
namespace {

template <int W>
__device__ __forceinline__ void bg_products(uint32_t x, const float* __restrict__ sV, float& p0, float& p1, float& p2,
                                            int max_k) {
  // digits x_0..x_{W-1}; BaMM ids are big-endian: (x_{i-2} x_{i-1} x_i)
  const uint32_t d0 = x & 3u;
  const float v0 = sV[d0];
  p0 = 1.0f * v0;
  p1 = p0;
  p2 = p0;
  uint32_t prev1 = d0;  // x_{i-1}
  uint32_t prev2 = 0;   // x_{i-2}
#pragma unroll
  for (int i = 1; i < W; ++i) {
    const uint32_t d = (x >> (2 * i)) & 3u;
    p0 *= sV[d];
    const float v1 = sV[4 + ((prev1 << 2) | d)];
    if (max_k >= 1) p1 *= v1;
    if (max_k >= 2) {
      if (i == 1)
        p2 *= v1;  // order-1 factor while only one base of context exists
      else
        p2 *= sV[20 + ((prev2 << 4) | (prev1 << 2) | d)];
    }
    prev2 = prev1;
    prev1 = d;
  }
}

// What the sweep leaves for one pattern from its aggregated probabilities and its count: expected, log-p, z
// (src/base_pattern.cpp:231-265; float / double semantics: the header of this file).
__device__ __forceinline__ void pattern_statistics(float pk, float fl, uint32_t n, float& mu, float& lp, float& zz) {
  mu = pk * fl;
  if (n == 0) {
    lp = __builtin_inff();
  } else {
    const float fn = (float)n;
    const float frac = (float)(1.0 - (double)(mu / (float)(n + 1u)));
    if (fn > mu && n > 5u) {
      const double dn = (double)n;
      lp = (float)(dn * log((double)(mu / fn)) + dn - (double)mu - 0.5 * log(6.283 * dn * (double)frac * (double)frac));
    } else {
      lp = 0.0f;
    }
  }
  zz = (float)((double)((float)n - mu) / sqrt((double)mu));
}

template <int W>
__global__ __launch_bounds__(256) void stats_kernel(int both, int k, int max_k, const float* __restrict__ V,
                                                    const unsigned long long* __restrict__ ltot_p,
                                                    const uint32_t* __restrict__ counts, float* __restrict__ bgprob,
                                                    float* __restrict__ expected, float* __restrict__ logp,
                                                    float* __restrict__ z) {
  __shared__ float sV[84];
  if (threadIdx.x < 84) sV[threadIdx.x] = V[threadIdx.x];
  __syncthreads();
  constexpr uint32_t NP = 1u << (2 * W);
  const float fl = (float)(*ltot_p);
  for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < NP; x += gridDim.x * blockDim.x) {
    float p[3];
    bg_products<W>(x, sV, p[0], p[1], p[2], max_k);
    if (both) {
      const uint32_t r = revcomp32(x, W);
      if (r != x) {
        float q[3];
        bg_products<W>(r, sV, q[0], q[1], q[2], max_k);
        // p[min] + p[max]: IEEE addition commutes, so operand order is irrelevant
        p[0] += q[0];
        p[1] += q[1];
        p[2] += q[2];
      }
    }
    bgprob[x] = p[0];
    if (max_k >= 1) bgprob[(size_t)NP + x] = p[1];
    if (max_k >= 2) bgprob[2 * (size_t)NP + x] = p[2];
    const float pk = k == 0 ? p[0] : (k == 1 ? p[1] : p[2]);
    float mu, lp, zz;
    pattern_statistics(pk, fl, counts[x], mu, lp, zz);
    expected[x] = mu;
    logp[x] = lp;
    z[x] = zz;
  }
}

// Both strands, W >= 12: a pattern and its reverse complement carry the same aggregated probabilities, the same expected
// count and -- in a mirrored count table, the only kind the callers hand over; any other is still handled -- the same
// count, log-p and z.  The one-thread-per-pattern kernel above works all of that out twice (492 vector instructions per
// pattern at W = 12, most of them the double-precision division, square root and logarithms: it is bound by its
// arithmetic at a quarter of the HBM rate).  Here a workgroup takes a TILE -- the 4096 patterns that share their middle
// W - 6 digits: 64 values of the three lowest digits x 64 of the three highest -- together with the tile of the reverse
// complements, which has the same shape (the twin of (lo, hi) in tile `mid` is (rc hi, rc lo) in tile rc(mid)): every
// pattern of the first tile is evaluated once, stored in place (256-byte runs: 64 consecutive lo), and handed to its twin
// through LDS, from where the second tile is stored in runs of the same length.  The 4^3 tiles that are their own twins
// are evaluated pattern by pattern as above.
constexpr uint32_t PAIR_THREADS = 1024, PAIR_TILE = 4096, PAIR_ROW = 65;  // (rows of 64 padded: the twin's place is the transposed one)
constexpr uint32_t PAIR_PER = PAIR_TILE / PAIR_THREADS;
// Two workgroups per CU (50 KB of LDS, 16 waves each), so that one's arithmetic runs beside the other's stores: a thread keeps
// its four patterns' results in registers, the twins' counts and then the twins' values go through ONE three-array buffer in
// two rounds (probabilities; expected / log-p / z).  (A first version staged all six arrays at once: 116 KB, one workgroup
// per CU, compute and store phases one after the other -- 0.35 ms at W = 12 against the per-pattern kernel's 0.236.)
template <int W>
__global__ __launch_bounds__(PAIR_THREADS) void stats_pair_kernel(const uint32_t* __restrict__ mids, uint32_t n_pair, int k, int max_k, const float* __restrict__ V,
                                                                const unsigned long long* __restrict__ ltot_p,
                                                                const uint32_t* __restrict__ counts, float* __restrict__ bgprob,
                                                                float* __restrict__ expected, float* __restrict__ logp,
                                                                float* __restrict__ z) {
  static_assert(W >= 8, "three low, three high digits and a middle");
  constexpr uint32_t NP = 1u << (2 * W);
  constexpr int MID = W - 6, HSHIFT = 2 * W - 6;
  __shared__ float sV[84];
  __shared__ float s_buf[3][64 * PAIR_ROW];
  uint32_t* s_cnt = reinterpret_cast<uint32_t*>(&s_buf[0][0]);  // (the twins' counts: read before the first value is staged)
  // (one workgroup per middle m <= rc(m), from a list: with workgroup = middle, half of them had nothing to do -- and which
  // half follows the middle's lowest digit, i.e. the XCD a workgroup lands on: two XCDs did half of the work)
  // The list: the n_pair middles m < rc(m), then the 4^((W-6)/2) middles that are their own twins.  A tile of the latter
  // needs no exchange, so it is cut into quarters -- PAIR_PER small workgroups, a pattern per thread -- that come LAST and
  // fill what the last round of whole tiles leaves of the chip (2016 tile pairs on 512 workgroup slots at W = 12: with
  // whole workgroups behind them a fifth round ran on an eighth of the CUs).
  const uint32_t t = threadIdx.x;
  const bool own_twin = blockIdx.x >= n_pair;
  const uint32_t entry = own_twin ? n_pair + (blockIdx.x - n_pair) / PAIR_PER : blockIdx.x;
  const uint32_t quarter = own_twin ? (blockIdx.x - n_pair) % PAIR_PER : 0u;
  const uint32_t mid = mids[entry], mid_tw = revcomp32(mid, MID);
  if (t < 84) sV[t] = V[t];
  const float fl = (float)(*ltot_p);
  auto id_of = [&](uint32_t m, uint32_t i) { return (i & 63u) | (m << 6) | ((i >> 6) << HSHIFT); };
  auto row_of = [](uint32_t i) { return (i & 63u) + PAIR_ROW * (i >> 6); };
  auto twin_at = [](uint32_t i) { return revcomp32(i >> 6, 3) + PAIR_ROW * revcomp32(i & 63u, 3); };  // (rc hi, rc lo) of the other tile
  if (!own_twin) {
#pragma unroll
    for (uint32_t j = 0; j < PAIR_PER; ++j) {
      const uint32_t i = j * PAIR_THREADS + t;
      s_cnt[row_of(i)] = counts[id_of(mid_tw, i)];
    }
  }
  __syncthreads();
  float p0[PAIR_PER], p1[PAIR_PER], p2[PAIR_PER], mu_tw[PAIR_PER], lp_tw[PAIR_PER], z_tw[PAIR_PER];
#pragma unroll
  for (uint32_t j = 0; j < PAIR_PER; ++j) {
    if (own_twin && j != 0u) break;
    const uint32_t i = (own_twin ? quarter : j) * PAIR_THREADS + t, x = id_of(mid, i);
    const uint32_t r = revcomp32(x, W);
    bg_products<W>(x, sV, p0[j], p1[j], p2[j], max_k);
    if (r != x) {
      float q[3];
      bg_products<W>(r, sV, q[0], q[1], q[2], max_k);
      p0[j] += q[0];
      p1[j] += q[1];
      p2[j] += q[2];
    }
    const float pk = k == 0 ? p0[j] : (k == 1 ? p1[j] : p2[j]);
    const uint32_t n = counts[x];
    float mu, lp, zz;
    pattern_statistics(pk, fl, n, mu, lp, zz);
    bgprob[x] = p0[j];
    if (max_k >= 1) bgprob[(size_t)NP + x] = p1[j];
    if (max_k >= 2) bgprob[2 * (size_t)NP + x] = p2[j];
    expected[x] = mu;
    logp[x] = lp;
    z[x] = zz;
    // the twin: same probabilities and expected count; log-p and z too when it has the same count
    mu_tw[j] = mu;
    lp_tw[j] = lp;
    z_tw[j] = zz;
    if (!own_twin) {
      const uint32_t n_tw = s_cnt[twin_at(i)];
      if (n_tw != n) pattern_statistics(pk, fl, n_tw, mu_tw[j], lp_tw[j], z_tw[j]);
    }
  }
  if (own_twin) return;
  __syncthreads();  // (everybody has read the counts)
#pragma unroll
  for (uint32_t j = 0; j < PAIR_PER; ++j) {
    const uint32_t at = twin_at(j * PAIR_THREADS + t);
    s_buf[0][at] = p0[j];
    s_buf[1][at] = p1[j];
    s_buf[2][at] = p2[j];
  }
  __syncthreads();
#pragma unroll
  for (uint32_t j = 0; j < PAIR_PER; ++j) {
    const uint32_t i = j * PAIR_THREADS + t, x = id_of(mid_tw, i), at = row_of(i);
    bgprob[x] = s_buf[0][at];
    if (max_k >= 1) bgprob[(size_t)NP + x] = s_buf[1][at];
    if (max_k >= 2) bgprob[2 * (size_t)NP + x] = s_buf[2][at];
  }
  __syncthreads();
#pragma unroll
  for (uint32_t j = 0; j < PAIR_PER; ++j) {
    const uint32_t at = twin_at(j * PAIR_THREADS + t);
    s_buf[0][at] = mu_tw[j];
    s_buf[1][at] = lp_tw[j];
    s_buf[2][at] = z_tw[j];
  }
  __syncthreads();
#pragma unroll
  for (uint32_t j = 0; j < PAIR_PER; ++j) {
    const uint32_t i = j * PAIR_THREADS + t, x = id_of(mid_tw, i), at = row_of(i);
    expected[x] = s_buf[0][at];
    logp[x] = s_buf[1][at];
    z[x] = s_buf[2][at];
  }
}

template <int W>
int launch_w(pengk_ctx* ctx, int both, int k, int max_k, const float* d_V, const uint64_t* d_ltot, const uint32_t* d_counts,
             float* d_bgprob, float* d_expected, float* d_logp, float* d_z) {
  if constexpr (W >= 12) {
    if (both && ctx->sweep_pairs) {
      constexpr uint32_t n_mid = 1u << (2 * (W - 6)), n_pal = 1u << (W - 6), n_wg = (n_mid + n_pal) / 2u;  // (4^(m/2) middles are their own twins)
      if (ctx->pair_mids_W != W) {
        std::vector<uint32_t> list;
        list.reserve(n_wg);
        for (uint32_t m = 0; m < n_mid; ++m)
          if (m < revcomp32(m, W - 6)) list.push_back(m);
        for (uint32_t m = 0; m < n_mid; ++m)
          if (m == revcomp32(m, W - 6)) list.push_back(m);
        if (list.size() != n_wg) return fail(PENGK_ERR_DEVICE, "twin-tile list: %zu middles, expected %u", list.size(), n_wg);
        int rc = ensure_scratch(ctx, &ctx->d_pair_mids, &ctx->pair_mids_bytes, list.size() * sizeof(uint32_t));
        if (rc) return rc;
        PENGK_HIP(hipMemcpyAsync(ctx->d_pair_mids, list.data(), list.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        PENGK_HIP(hipStreamSynchronize(ctx->stream));  // (the list leaves scope; once per context and pattern length)
        ctx->pair_mids_W = W;
      }
      hipLaunchKernelGGL((stats_pair_kernel<W>), dim3(n_wg - n_pal + n_pal * PAIR_PER), dim3(PAIR_THREADS), 0, ctx->stream, (const uint32_t*)ctx->d_pair_mids, n_wg - n_pal, k, max_k, d_V,
                         (const unsigned long long*)d_ltot, d_counts, d_bgprob, d_expected, d_logp, d_z);
      PENGK_HIP(hipGetLastError());
      return PENGK_OK;
    }
  }
  const uint32_t np = 1u << (2 * W);
  const uint32_t need = (np + 255) / 256;
  const uint32_t cap = (uint32_t)ctx->num_cu * 16u;
  hipLaunchKernelGGL((stats_kernel<W>), dim3(need < cap ? need : cap), dim3(256), 0, ctx->stream, both, k, max_k, d_V,
                     (const unsigned long long*)d_ltot, d_counts, d_bgprob, d_expected, d_logp, d_z);
  PENGK_HIP(hipGetLastError());
  return PENGK_OK;
}

// ---- seed candidates (BasePattern::select_base_patterns, src/base_pattern.cpp:443-515, first half) -----------------
// The reference sorts ALL 4^W ids by z and walks the ranking down to the threshold; only ids with z >= threshold and
// count >= threshold can become seeds.  This kernel compacts exactly those (id, z) pairs: wave ballot, one atomic per
// wave.  Order of the list is arbitrary; the caller sorts the few thousand survivors.
__global__ __launch_bounds__(256) void seed_candidates_kernel(const float* __restrict__ z, const uint32_t* __restrict__ counts,
                                                              uint32_t np, float z_threshold, uint32_t count_threshold,
                                                              uint32_t cap, uint32_t* __restrict__ n_out,
                                                              uint32_t* __restrict__ ids, float* __restrict__ zs) {
  const uint32_t lane = threadIdx.x & 63u;
  for (uint32_t base = (blockIdx.x * blockDim.x + threadIdx.x) & ~63u; base < np; base += gridDim.x * blockDim.x) {
    const uint32_t x = base + lane;
    const float zx = x < np ? z[x] : 0.0f;
    const bool keep = x < np && !(zx < z_threshold) && counts[x] >= count_threshold;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
    if (m == 0) continue;
    uint32_t first = 0;
    if (lane == 0) first = atomicAdd(n_out, (uint32_t)__builtin_popcountll(m));
    first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
    if (keep) {
      const uint32_t at = first + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
      if (at < cap) {
        ids[at] = x;
        zs[at] = zx;
      }
    }
  }
}

}  // namespace

int launch_seed_candidates(pengk_ctx* ctx, int W, const float* d_z, const uint32_t* d_counts, float z_threshold,
                           uint32_t count_threshold, uint32_t cap, uint32_t* d_n, uint32_t* d_ids, float* d_zs) {
  const uint32_t np = 1u << (2 * W);
  PENGK_HIP(hipMemsetAsync(d_n, 0, sizeof(uint32_t), ctx->stream));
  const uint32_t need = (np + 255) / 256, lim = (uint32_t)ctx->num_cu * 8u;
  hipLaunchKernelGGL(seed_candidates_kernel, dim3(need < lim ? need : lim), dim3(256), 0, ctx->stream, d_z, d_counts, np,
                     z_threshold, count_threshold, cap, d_n, d_ids, d_zs);
  PENGK_HIP(hipGetLastError());
  return PENGK_OK;
}

int launch_stats(pengk_ctx* ctx, int W, int both, int k, int max_k, const float* d_V, const uint64_t* d_ltot,
                 const uint32_t* d_counts, float* d_bgprob, float* d_expected, float* d_logp, float* d_z) {
  switch (W) {
    case 2: return launch_w<2>(ctx, both, k, max_k, d_V, d_ltot, d_counts, d_bgprob, d_expected, d_logp, d_z);
    case 4: return launch_w<4>(ctx, both, k, max_k, d_V, d_ltot, d_counts, d_bgprob, d_expected, d_logp, d_z);
    case 6: return launch_w<6>(ctx, both, k, max_k, d_V, d_ltot, d_counts, d_bgprob, d_expected, d_logp, d_z);
    case 8: return launch_w<8>(ctx, both, k, max_k, d_V, d_ltot, d_counts, d_bgprob, d_expected, d_logp, d_z);
    case 10: return launch_w<10>(ctx, both, k, max_k, d_V, d_ltot, d_counts, d_bgprob, d_expected, d_logp, d_z);
    case 12: return launch_w<12>(ctx, both, k, max_k, d_V, d_ltot, d_counts, d_bgprob, d_expected, d_logp, d_z);
    case 14: return launch_w<14>(ctx, both, k, max_k, d_V, d_ltot, d_counts, d_bgprob, d_expected, d_logp, d_z);
    default: return fail(PENGK_ERR_ARG, "pattern length %d unsupported", W);
  }
}

// (pengk_warmup: loads this translation unit's code object ahead of its first launch)
int warm_stats() {
  hipFuncAttributes a;
  PENGK_HIP(hipFuncGetAttributes(&a, (const void*)seed_candidates_kernel));
  return PENGK_OK;
}

}  // namespace pengk
