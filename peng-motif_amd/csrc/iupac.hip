// iupac.hip -- K4: IUPAC degenerate-pattern aggregation, one thread per pattern (gfx950).
//
// Replaces IUPACPattern::basepatterns_from_iupac_{single,double}_stranded + the summation loop of
// aggregate_attributes_from_basepatterns / count_combined_occurences
// (src/iupac_pattern.cpp:331-473,806-833).  No id list is materialised and nothing is sorted:
//
//  * PLUS: the reference's LIFO stack emits an odometer in which the LAST degenerate position moves
//    fastest and every position runs rep[0], rep[n-1], ..., rep[1]; the thread walks that odometer
//    and adds count / background / expected in exactly that order (the float32 sums are
//    order-sensitive).
//  * BOTH: the reference canonicalises (min(id, rc)), sorts ascending and skips adjacent
//    duplicates, i.e. it sums over the DISTINCT canonical ids in ascending order.  That set is
//    { c in E(P) u E(rc P) : c <= rc(c) }, so the thread merges two ascending odometers (P and its
//    reverse-complement pattern), keeps canonical members and drops the duplicate when both
//    streams carry the same id.
//
// The three sums come back to the host; z-score and log-p (O(1) scalar double math per pattern,
// :446-470) are finished in launch_iupac with the same libm the reference links.
#include <math.h>

#include <limits>
#include <new>

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

// ascending odometer over the expansion of one IUPAC pattern (position W-1 most significant)
struct AscStream {
  int letter[MAXW];
  int idx[MAXW];
  uint32_t cur;
  bool done;
  int W;
  __device__ void init(const int* letters, int W_) {
    W = W_;
    cur = 0;
    done = false;
    for (int p = 0; p < W; ++p) {
      letter[p] = letters[p];
      idx[p] = 0;
      cur |= (uint32_t)c_rep[letters[p]][0] << (2 * p);
    }
  }
  __device__ void step() {
    for (int p = 0; p < W; ++p) {
      const int L = letter[p];
      const int n = c_rep_n[L];
      if (idx[p] + 1 < n) {
        cur += (uint32_t)(c_rep[L][idx[p] + 1] - c_rep[L][idx[p]]) << (2 * p);
        ++idx[p];
        return;
      }
      cur -= (uint32_t)(c_rep[L][idx[p]] - c_rep[L][0]) << (2 * p);
      idx[p] = 0;
    }
    done = true;
  }
  // move to the first canonical member at or after the current one
  __device__ void settle() {
    while (!done && cur > revcomp32(cur, W)) step();
  }
};

__global__ __launch_bounds__(64) void iupac_kernel(int W, int both, const unsigned long long* __restrict__ ids, int n,
                                                   const uint32_t* __restrict__ counts, const float* __restrict__ bgp,
                                                   const float* __restrict__ expected, RawSums* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int letters[MAXW];
  {
    unsigned long long t = ids[i];
    for (int p = 0; p < W; ++p) {
      letters[p] = (int)(t % 11ull);
      t /= 11ull;
    }
  }
  unsigned long long sum_c = 0;
  float sum_bg = 0.0f, sum_e = 0.0f;
  if (!both) {
    int stepv[MAXW];
    unsigned long long total = 1;
    for (int p = 0; p < W; ++p) {
      stepv[p] = 0;
      total *= (unsigned long long)c_rep_n[letters[p]];
    }
    for (unsigned long long k = 0; k < total; ++k) {
      uint32_t x = 0;
      for (int p = 0; p < W; ++p) {
        const int cnt = c_rep_n[letters[p]];
        const int j = stepv[p] == 0 ? 0 : cnt - stepv[p];
        x |= (uint32_t)c_rep[letters[p]][j] << (2 * p);
      }
      sum_bg += bgp[x];
      sum_c += counts[x];
      sum_e += expected[x];
      for (int p = W - 1; p >= 0; --p) {
        if (++stepv[p] < c_rep_n[letters[p]]) break;
        stepv[p] = 0;
      }
    }
  } else {
    int rcl[MAXW];
    for (int p = 0; p < W; ++p) rcl[p] = c_comp[letters[W - 1 - p]];
    AscStream a, b;
    a.init(letters, W);
    b.init(rcl, W);
    a.settle();
    b.settle();
    while (!a.done || !b.done) {
      const uint32_t xa = a.done ? 0xFFFFFFFFu : a.cur;
      const uint32_t xb = b.done ? 0xFFFFFFFFu : b.cur;
      const uint32_t x = xa < xb ? xa : xb;
      sum_bg += bgp[x];
      sum_c += counts[x];
      sum_e += expected[x];
      if (xa == x) {
        a.step();
        a.settle();
      }
      if (xb == x) {
        b.step();
        b.settle();
      }
    }
  }
  out[i].sites = sum_c;
  out[i].bg_p = sum_bg;
  out[i].expected = sum_e;
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
  int rc = ensure_scratch(ctx, &ctx->d_misc, &ctx->misc_bytes, id_pad + out_bytes);
  if (rc) return rc;
  unsigned long long* d_ids = (unsigned long long*)ctx->d_misc;
  RawSums* d_out = (RawSums*)((char*)ctx->d_misc + id_pad);
  PENGK_HIP(hipMemcpyAsync(d_ids, h_ids, id_bytes, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(iupac_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, W, both, d_ids, (int)n, d_counts,
                     d_bgp, d_expected, d_out);
  PENGK_HIP(hipGetLastError());
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

}  // namespace pengk
