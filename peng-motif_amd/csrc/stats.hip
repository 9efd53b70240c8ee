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
    const float mu = pk * fl;
    expected[x] = mu;
    const uint32_t n = counts[x];
    float lp;
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
    logp[x] = lp;
    z[x] = (float)((double)((float)n - mu) / sqrt((double)mu));
  }
}

template <int W>
int launch_w(pengk_ctx* ctx, int both, int k, int max_k, const float* d_V, const uint64_t* d_ltot, const uint32_t* d_counts,
             float* d_bgprob, float* d_expected, float* d_logp, float* d_z) {
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
