// em.hip -- K5: EM of every PWM against the whole 4^W k-mer table (gfx950).
//
// Replaces Peng::em_optimize_pwms + Peng::calculate_prob_odds (src/peng.cpp:48-197) and the row
// normalisation IUPACPattern::normalize_pwm (src/iupac_pattern.cpp:291-303).
//
// One launch handles one EM iteration of a whole batch of PWMs: grid = (blocks per PWM, PWMs).
// Pattern id x = [hi | mid | lo]: the 4 low digits are the thread index (so a wave reads
// consecutive count / background entries), the `mid` digits are the block index, and the HI (<= 4)
// top digits are walked by the thread in a depth-first loop nest that re-uses the partial products
// exactly like the reference's recursion does -- the float32 product is built in position order
// 0..W-1, so odds[x] and the per-k-mer weight  c*s / (1 + s/odds)  carry the reference's bits.
// The PWM columns are staged in LDS once per block.
//
// What differs from the reference is only the summation of the 4^W weights per PWM cell: the
// reference adds them serially in float32 (error up to 2.6e-4 relative at W=10);
// here they are accumulated in fp64 through a fixed tree (thread -> wave -> block -> grid), so the
// result is deterministic and within 1 ulp(float) of the exact sum.
#include "pengk_internal.h"

namespace pengk {
namespace {

// HIMAX = 4: 256 leaves per thread (fewest partial products; best when the grid is full anyway).
// HIMAX = 2: 16 leaves per thread, 16x more workgroups -- for small PWM batches that would otherwise leave
// most CUs with a single wave (the 16-PWM batch of a typical run).
template <int W, int HIMAX = 4>
struct EmGeo {
  static constexpr int LO = 4;                               // digits taken from threadIdx (256 threads)
  static constexpr int HI = (W - LO) < HIMAX ? (W - LO) : HIMAX;  // digits walked per thread
  static constexpr int MID = W - LO - HI;                    // digits taken from blockIdx.x
  static constexpr int PB = LO + MID;                        // first HI position
  static constexpr int NB = 1 << (2 * MID);                  // blocks per PWM
  static constexpr int CELLS = W * 4;
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;  // lane 0 holds the sum; fixed tree => deterministic
}

// fast mode: the two per-k-mer factors that do not depend on the PWM, once per call instead of once per PWM and iteration
__global__ __launch_bounds__(256) void em_prepare_kernel(const uint32_t* __restrict__ counts, const float* __restrict__ bg,
                                                         float saturation, uint32_t np, float* __restrict__ cs,
                                                         float* __restrict__ sb) {
  for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < np; x += gridDim.x * blockDim.x) {
    cs[x] = (float)counts[x] * saturation;
    sb[x] = saturation * bg[x];
  }
}

// FAST: `counts` / `bg` are reinterpreted as the float tables of em_prepare_kernel (cs, sb)
template <int W, int HIMAX, bool FAST>
__global__ __launch_bounds__(256) void em_accumulate_kernel(const float* __restrict__ pwms, const int32_t* __restrict__ state,
                                                            const uint32_t* __restrict__ counts,
                                                            const float* __restrict__ bg, float saturation,
                                                            double* __restrict__ partials) {
  using G = EmGeo<W, HIMAX>;
  const int pw = blockIdx.y;
  if (state[2 * pw + 1] == 0) return;  // converged or out of iterations (block-uniform)

  __shared__ float s_pwm[W * 4];
  __shared__ double s_red[4][G::HI > 0 ? G::HI * 4 : 1];
  if (threadIdx.x < W * 4) s_pwm[threadIdx.x] = pwms[(size_t)pw * W * 4 + threadIdx.x];
  __syncthreads();

  const uint32_t tid = threadIdx.x;
  const uint32_t mid = blockIdx.x;
  const uint32_t xlow = tid | (mid << (2 * G::LO));

  // prefix product over positions 0 .. PB-1 (reference order: ((1*p0)*p1)*...)
  float pr = 1.0f;
#pragma unroll
  for (int p = 0; p < G::PB; ++p) pr = pr * s_pwm[p * 4 + ((xlow >> (2 * p)) & 3u)];

  double acc[G::HI > 0 ? G::HI : 1][4];
#pragma unroll
  for (int h = 0; h < (G::HI > 0 ? G::HI : 1); ++h)
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[h][a] = 0.0;
  double S = 0.0;

  auto leaf = [&](uint32_t x, float prod) -> double {
    if constexpr (FAST) {
      // c*s / (1 + s/(prod/bg)) == c*s*prod / (prod + s*bg): one reciprocal (v_rcp_f32, 1 ulp) instead of three
      // IEEE divisions (33 of the 42 VALU instructions of a leaf); c*s and s*bg come precomputed.
      // Same limits: prod = 0 -> 0.
      const float cs = __builtin_bit_cast(float, counts[x]);
      const float den = bg[x] + prod;
      return (double)(cs * prod * __builtin_amdgcn_rcpf(den));
    } else {
      const float odds = prod / bg[x];
      const float w = ((float)counts[x] * saturation) / (1 + saturation / odds);  // src/peng.cpp:124-125
      return (double)w;
    }
  };

  if constexpr (G::HI == 0) {
    S = leaf(xlow, pr);
  } else if constexpr (G::HI == 2) {
#pragma unroll
    for (int d0 = 0; d0 < 4; ++d0) {
      const float p0 = pr * s_pwm[(G::PB + 0) * 4 + d0];
      double s0 = 0.0;
#pragma unroll
      for (int d1 = 0; d1 < 4; ++d1) {
        const float p1 = p0 * s_pwm[(G::PB + 1) * 4 + d1];
        const uint32_t x = xlow | ((uint32_t)d0 << (2 * G::PB)) | ((uint32_t)d1 << (2 * (G::PB + 1)));
        const double w = leaf(x, p1);
        acc[1][d1] += w;
        s0 += w;
      }
      acc[0][d0] += s0;
      S += s0;
    }
  } else {
    static_assert(G::HI == 4 || G::HI == 0 || G::HI == 2, "EM geometry");
    // the two outer digits stay loops: unrolled 256 leaves deep the kernel needs > 256 VGPRs (one wave per SIMD)
#pragma unroll 1
    for (int d0 = 0; d0 < 4; ++d0) {
      const float p0 = pr * s_pwm[(G::PB + 0) * 4 + d0];
      double s0 = 0.0;
#pragma unroll 1
      for (int d1 = 0; d1 < 4; ++d1) {
        const float p1 = p0 * s_pwm[(G::PB + 1) * 4 + d1];
        double s1 = 0.0;
#pragma unroll
        for (int d2 = 0; d2 < 4; ++d2) {
          const float p2 = p1 * s_pwm[(G::PB + 2) * 4 + d2];
          double s2 = 0.0;
#pragma unroll
          for (int d3 = 0; d3 < 4; ++d3) {
            const float p3 = p2 * s_pwm[(G::PB + 3) * 4 + d3];
            const uint32_t x = xlow | ((uint32_t)d0 << (2 * G::PB)) | ((uint32_t)d1 << (2 * (G::PB + 1))) |
                               ((uint32_t)d2 << (2 * (G::PB + 2))) | ((uint32_t)d3 << (2 * (G::PB + 3)));
            const double w = leaf(x, p3);
            acc[3][d3] += w;
            s2 += w;
          }
          acc[2][d2] += s2;
          s1 += s2;
        }
        acc[1][d1] += s1;
        s0 += s1;
      }
      acc[0][d0] += s0;
      S += s0;
    }
  }

  // ---- block reduction (fixed tree) --------------------------------------------------------------
  // Digits 0..2 of the pattern id are lane bits, digit 3 is the wave index.  Cell (p, a) for p < 3 is the sum of
  // S over the lanes whose digit p equals a: an xor butterfly over the four lane bits outside digit p leaves it
  // in every lane of the class; lane a << 2p publishes it.  (A serial pass over 256 LDS values per cell was most
  // of the kernel for small PWM batches.)
  const int wave = tid >> 6, lane = tid & 63;
  __shared__ double s_cls[4][3][4];
  __shared__ double s_T[4];
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    double v = S;
#pragma unroll
    for (int b = 0; b < 6; ++b)
      if ((b >> 1) != p) v += __shfl_xor(v, 1 << b, 64);
    if ((lane & ~(3 << (2 * p))) == 0) s_cls[wave][p][(lane >> (2 * p)) & 3] = v;
    if (p == 0) {  // whole wave
      double t = v;
      t += __shfl_xor(t, 1, 64);
      t += __shfl_xor(t, 2, 64);
      if (lane == 0) s_T[wave] = t;
    }
  }
  if constexpr (G::HI > 0) {
#pragma unroll
    for (int h = 0; h < G::HI; ++h)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const double v = wave_sum(acc[h][a]);
        if (lane == 0) s_red[wave][h * 4 + a] = v;
      }
  }
  __syncthreads();
  double* out = partials + ((size_t)pw * G::NB + blockIdx.x) * G::CELLS;
  if (tid < (uint32_t)G::CELLS) {
    const int p = tid >> 2, a = tid & 3;
    double v = 0.0;
    if (p < 3) {
      v = ((s_cls[0][p][a] + s_cls[1][p][a]) + s_cls[2][p][a]) + s_cls[3][p][a];
    } else if (p == 3) {
      v = s_T[a];
    } else if (p < G::PB) {  // digit fixed by the block index
      if ((int)((mid >> (2 * (p - G::LO))) & 3u) == a) v = ((s_T[0] + s_T[1]) + s_T[2]) + s_T[3];
    } else {
      if constexpr (G::HI > 0) {
        const int h = p - G::PB;
        v = ((s_red[0][h * 4 + a] + s_red[1][h * 4 + a]) + s_red[2][h * 4 + a]) + s_red[3][h * 4 + a];
      }
    }
    out[tid] = v;
  }
}

// One block per PWM: sum the per-block partials in block order, then the reference's float32
// epilogue: normalise rows (:129), change = sum |new - old| (:132-137), swap (:140-143).
template <int W, int HIMAX>
__global__ __launch_bounds__(64) void em_finalize_kernel(float* __restrict__ pwms, int32_t* __restrict__ state,
                                                         float* __restrict__ change_out, const double* __restrict__ partials,
                                                         float threshold, int max_it) {
  using G = EmGeo<W, HIMAX>;
  const int pw = blockIdx.x;
  if (state[2 * pw + 1] == 0) return;
  __shared__ float s_new[W * 4];
  const int e = threadIdx.x;
  if (e < G::CELLS) {
    const double* src = partials + (size_t)pw * G::NB * G::CELLS + e;
    double v = 0.0;
    for (int b = 0; b < G::NB; ++b) v += src[(size_t)b * G::CELLS];
    s_new[e] = (float)v;
  }
  __syncthreads();
  if (e == 0) {
    float* old = pwms + (size_t)pw * W * 4;
    float change = 0.0f;
    for (int p = 0; p < W; ++p) {
      float sum = 0.0f;
      for (int a = 0; a < 4; ++a) sum += s_new[p * 4 + a];
      for (int a = 0; a < 4; ++a) s_new[p * 4 + a] /= sum;
    }
    for (int p = 0; p < W; ++p)
      for (int a = 0; a < 4; ++a) {
        change += fabsf(s_new[p * 4 + a] - old[p * 4 + a]);
        old[p * 4 + a] = s_new[p * 4 + a];
      }
    const int it = state[2 * pw] + 1;
    state[2 * pw] = it;
    state[2 * pw + 1] = !(change <= threshold || it >= max_it);
    change_out[pw] = change;
  }
}

__global__ void em_init_kernel(int n, int W, float threshold, int max_it, int32_t* __restrict__ state,
                               float* __restrict__ change) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float c0 = (float)W;  // `float change = pattern_length` (src/peng.cpp:101)
  state[2 * i] = 0;
  state[2 * i + 1] = !(c0 <= threshold || 0 >= max_it);
  change[i] = c0;
}

template <int W, int HIMAX, bool FAST>
int launch_geo(pengk_ctx* ctx, int64_t n_pwm, float* d_pwms, float saturation, float threshold, int max_it,
               const uint32_t* d_counts, const float* d_bg, int32_t* d_state, float* d_change) {
  using G = EmGeo<W, HIMAX>;
  hipLaunchKernelGGL(em_init_kernel, dim3((unsigned)((n_pwm + 255) / 256)), dim3(256), 0, ctx->stream, (int)n_pwm, W, threshold,
                     max_it, d_state, d_change);
  PENGK_HIP(hipGetLastError());
  const size_t per_pwm = (size_t)G::NB * G::CELLS * sizeof(double);
  const size_t budget = (size_t)256 << 20;
  int64_t batch = (int64_t)(budget / per_pwm);
  if (batch < 1) batch = 1;
  if (batch > n_pwm) batch = n_pwm;
  if (batch > 65535) batch = 65535;  // gridDim.y
  int rc = ensure_scratch(ctx, (void**)&ctx->d_em_partials, &ctx->em_partials_bytes, (size_t)batch * per_pwm);
  if (rc) return rc;
  if (FAST) {
    const uint32_t np = 1u << (2 * W);
    rc = ensure_scratch(ctx, (void**)&ctx->d_em_tables, &ctx->em_tables_bytes, (size_t)2 * np * sizeof(float));
    if (rc) return rc;
    float* cs = ctx->d_em_tables;
    float* sb = cs + np;
    const unsigned pb = (np + 255) / 256 < 4096u ? (np + 255) / 256 : 4096u;
    hipLaunchKernelGGL(em_prepare_kernel, dim3(pb), dim3(256), 0, ctx->stream, d_counts, d_bg, saturation, np, cs, sb);
    PENGK_HIP(hipGetLastError());
    d_counts = reinterpret_cast<const uint32_t*>(cs);
    d_bg = sb;
  }
  for (int64_t first = 0; first < n_pwm; first += batch) {
    const int64_t nb = n_pwm - first < batch ? n_pwm - first : batch;
    for (int it = 0; it < max_it; ++it) {
      hipLaunchKernelGGL((em_accumulate_kernel<W, HIMAX, FAST>), dim3(G::NB, (unsigned)nb), dim3(256), 0, ctx->stream,
                         d_pwms + (size_t)first * W * 4, d_state + 2 * first, d_counts, d_bg, saturation, ctx->d_em_partials);
      hipLaunchKernelGGL((em_finalize_kernel<W, HIMAX>), dim3((unsigned)nb), dim3(64), 0, ctx->stream,
                         d_pwms + (size_t)first * W * 4, d_state + 2 * first, d_change + first, ctx->d_em_partials, threshold, max_it);
    }
    PENGK_HIP(hipGetLastError());
  }
  return PENGK_OK;
}

template <int W>
int launch_w(pengk_ctx* ctx, int64_t n_pwm, float* d_pwms, float saturation, float threshold, int max_it,
             const uint32_t* d_counts, const float* d_bg, int32_t* d_state, float* d_change) {
  // few PWMs: more, smaller workgroups so that every CU gets several waves
  const bool fast = ctx->em_fast != 0;
  if constexpr (W >= 8) {
    const int64_t wg4 = n_pwm * EmGeo<W, 4>::NB;
    if (wg4 < (int64_t)ctx->num_cu * 8)
      return fast ? launch_geo<W, 2, true>(ctx, n_pwm, d_pwms, saturation, threshold, max_it, d_counts, d_bg, d_state, d_change)
                  : launch_geo<W, 2, false>(ctx, n_pwm, d_pwms, saturation, threshold, max_it, d_counts, d_bg, d_state, d_change);
  }
  return fast ? launch_geo<W, 4, true>(ctx, n_pwm, d_pwms, saturation, threshold, max_it, d_counts, d_bg, d_state, d_change)
              : launch_geo<W, 4, false>(ctx, n_pwm, d_pwms, saturation, threshold, max_it, d_counts, d_bg, d_state, d_change);
}

}  // namespace

int launch_em(pengk_ctx* ctx, int W, int64_t n_pwm, float* d_pwms, float saturation, float threshold, int max_it,
              const uint32_t* d_counts, const float* d_bg, int32_t* d_state, float* d_change) {
  switch (W) {
    case 4: return launch_w<4>(ctx, n_pwm, d_pwms, saturation, threshold, max_it, d_counts, d_bg, d_state, d_change);
    case 6: return launch_w<6>(ctx, n_pwm, d_pwms, saturation, threshold, max_it, d_counts, d_bg, d_state, d_change);
    case 8: return launch_w<8>(ctx, n_pwm, d_pwms, saturation, threshold, max_it, d_counts, d_bg, d_state, d_change);
    case 10: return launch_w<10>(ctx, n_pwm, d_pwms, saturation, threshold, max_it, d_counts, d_bg, d_state, d_change);
    case 12: return launch_w<12>(ctx, n_pwm, d_pwms, saturation, threshold, max_it, d_counts, d_bg, d_state, d_change);
    case 14: return launch_w<14>(ctx, n_pwm, d_pwms, saturation, threshold, max_it, d_counts, d_bg, d_state, d_change);
    default: return fail(PENGK_ERR_ARG, "pattern length %d unsupported", W);
  }
}

}  // namespace pengk
