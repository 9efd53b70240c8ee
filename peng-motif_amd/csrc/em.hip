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
#include "em_serial.h"

namespace pengk {
namespace {
// Developer build -DPENGK_WG_TRACE (tools/em_wgtrace.py): every workgroup of the three kernels of the blocks-ahead EM leaves
// {kernel, kind, workgroup, XCC, start, end} (s_memrealtime: the 100 MHz clock all XCCs share; s_memtime has a base of its own per XCC) in a device array -- where a kernel's microseconds go
// when its instruction count explains a third of them.  Never part of the product build.
#ifdef PENGK_WG_TRACE
constexpr unsigned WG_TRACE_MAX = 1u << 20;
__device__ unsigned long long g_wg_trace[3 * WG_TRACE_MAX];
__device__ unsigned g_wg_trace_n;
struct WgTrace {
  unsigned long long t0;
  unsigned kernel;
  __device__ __forceinline__ WgTrace(unsigned k) : t0(__builtin_amdgcn_s_memrealtime()), kernel(k) {}
  __device__ __forceinline__ void end(unsigned kind, unsigned wg) const {
    if (threadIdx.x != 0) return;
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 << 11) | (0 << 6) | 20) & 0xFu;  // HW_REG_XCC_ID, bits 3:0
    const unsigned i = atomicAdd(&g_wg_trace_n, 1u);
    if (i < WG_TRACE_MAX) {
      g_wg_trace[3 * i] = ((unsigned long long)kernel << 56) | ((unsigned long long)kind << 48) | ((unsigned long long)xcc << 40) | wg;
      g_wg_trace[3 * i + 1] = t0;
      g_wg_trace[3 * i + 2] = t1;
    }
  }
};
#define PENGK_WG_TRACE_BEGIN(k) const WgTrace wg_trace(k)
#define PENGK_WG_TRACE_END(kind, wg) wg_trace.end(kind, wg)
#else
#define PENGK_WG_TRACE_BEGIN(k)
#define PENGK_WG_TRACE_END(kind, wg)
#endif

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

// FAST: `counts` / `bg` are reinterpreted as the float tables of em_prepare_kernel (cs, sb).
// P PWMs per workgroup: a thread evaluates its k-mers for P PWMs with ONE read of the two table entries (the
// table comes from L2 / Infinity Cache once per PWM otherwise).  Only P = 1 is dispatched, see launch_w.
template <int W, int HIMAX, bool FAST, int P>
__global__ __launch_bounds__(256) void em_accumulate_kernel(const float* __restrict__ pwms, const int32_t* __restrict__ state,
                                                            const uint32_t* __restrict__ counts,
                                                            const float* __restrict__ bg, float saturation,
                                                            double* __restrict__ partials, int n_pwm) {
  using G = EmGeo<W, HIMAX>;
  constexpr int HIN = G::HI > 0 ? G::HI : 1;
  const int pw0 = blockIdx.y * P;
  bool live[P];
  bool any = false;
#pragma unroll
  for (int q = 0; q < P; ++q) {
    live[q] = pw0 + q < n_pwm && state[2 * (pw0 + q) + 1] != 0;  // not converged, iterations left (block-uniform)
    any |= live[q];
  }
  if (!any) return;

  __shared__ float s_pwm[P][W * 4];
  __shared__ double s_red[P][4][HIN * 4];
  __shared__ double s_cls[P][4][3][4];
  __shared__ double s_T[P][4];
  for (int i = threadIdx.x; i < P * W * 4; i += blockDim.x) {
    const int q = i / (W * 4), c = i % (W * 4);
    const int src = pw0 + q < n_pwm ? pw0 + q : n_pwm - 1;  // a PWM past the end is computed and dropped
    s_pwm[q][c] = pwms[(size_t)src * W * 4 + c];
  }
  __syncthreads();

  const uint32_t tid = threadIdx.x;
  const uint32_t mid = blockIdx.x;
  const uint32_t xlow = tid | (mid << (2 * G::LO));

  // prefix product over positions 0 .. PB-1 (reference order: ((1*p0)*p1)*...)
  float pr[P];
  double acc[P][HIN][4];
  double S[P];
#pragma unroll
  for (int q = 0; q < P; ++q) {
    pr[q] = 1.0f;
#pragma unroll
    for (int p = 0; p < G::PB; ++p) pr[q] = pr[q] * s_pwm[q][p * 4 + ((xlow >> (2 * p)) & 3u)];
#pragma unroll
    for (int h = 0; h < HIN; ++h)
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[q][h][a] = 0.0;
    S[q] = 0.0;
  }

  // weights of k-mer x for the P products
  auto leaf = [&](uint32_t x, const float (&prod)[P], double (&w)[P]) {
    if constexpr (FAST) {
      // c*s / (1 + s/(prod/bg)) == c*s*prod / (prod + s*bg): one reciprocal (v_rcp_f32, 1 ulp) instead of three
      // IEEE divisions (33 of the 42 VALU instructions of a leaf); c*s and s*bg come precomputed.
      // Same limits: prod = 0 -> 0.
      // 32-bit byte offset + uniform base: one address instruction per k-mer for both tables (indexing the two
      // pointers with x costs five 64-bit address instructions per k-mer, a third of what this kernel issued)
      const uint32_t off = x << 2;
      const float cs = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(counts) + off);
      const float sb = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(bg) + off);
#pragma unroll
      for (int q = 0; q < P; ++q) w[q] = (double)(cs * prod[q] * __builtin_amdgcn_rcpf(sb + prod[q]));
    } else {
      const uint32_t off = x << 2;
      const float cs = (float)*reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(counts) + off) * saturation;
      const float b = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(bg) + off);
#pragma unroll
      for (int q = 0; q < P; ++q) {
        const float odds = prod[q] / b;
        w[q] = (double)(cs / (1 + saturation / odds));  // src/peng.cpp:124-125
      }
    }
  };

  if constexpr (G::HI == 0) {
    leaf(xlow, pr, S);
  } else {
    static_assert(G::HI >= 2, "EM geometry");
    // HI digits per thread: the outer HI - 2 are walked by a run-time loop (unrolled 256 leaves deep the kernel
    // needs > 256 VGPRs: one wave per SIMD), the inner two are unrolled: blocks of 16 leaves.
    constexpr int OUT = G::HI - 2;
    double so[P][OUT > 0 ? OUT : 1];  // running sums of the outer digits' cells
#pragma unroll
    for (int q = 0; q < P; ++q)
#pragma unroll
      for (int jo = 0; jo < (OUT > 0 ? OUT : 1); ++jo) so[q][jo] = 0.0;
#pragma unroll 1
    for (int o = 0; o < (1 << (2 * OUT)); ++o) {
      float po[P];  // product up to the last outer position, in position order like the reference's recursion
      uint32_t xo = xlow;
#pragma unroll
      for (int q = 0; q < P; ++q) po[q] = pr[q];
#pragma unroll
      for (int jo = 0; jo < OUT; ++jo) {
        const int dj = (o >> (2 * jo)) & 3;
        xo |= (uint32_t)dj << (2 * (G::PB + jo));
#pragma unroll
        for (int q = 0; q < P; ++q) po[q] = po[q] * s_pwm[q][(G::PB + jo) * 4 + dj];
      }
      double s1[P];
#pragma unroll
      for (int q = 0; q < P; ++q) s1[q] = 0.0;
      if constexpr (FAST) {
        // throughput mode: the 16 weights of a block are summed by rows and columns in float32 (sums of 4: within
        // 2e-7 of exact, far inside this mode's 1e-5) and enter the fp64 tree as 9 values instead of 32 fp64 additions
        float wf[P][4][4];
#pragma unroll
        for (int d2 = 0; d2 < 4; ++d2) {
          float p2[P];
#pragma unroll
          for (int q = 0; q < P; ++q) p2[q] = po[q] * s_pwm[q][(G::PB + OUT) * 4 + d2];
#pragma unroll
          for (int d3 = 0; d3 < 4; ++d3) {
            const uint32_t off = (xo | ((uint32_t)d2 << (2 * (G::PB + OUT))) | ((uint32_t)d3 << (2 * (G::PB + OUT + 1)))) << 2;
            const float cs = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(counts) + off);
            const float sb = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(bg) + off);
#pragma unroll
            for (int q = 0; q < P; ++q) {
              const float p3 = p2[q] * s_pwm[q][(G::PB + OUT + 1) * 4 + d3];
              wf[q][d2][d3] = cs * p3 * __builtin_amdgcn_rcpf(sb + p3);
            }
          }
        }
#pragma unroll
        for (int q = 0; q < P; ++q) {
          float tot = 0.0f;
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const float row = (wf[q][d][0] + wf[q][d][1]) + (wf[q][d][2] + wf[q][d][3]);
            const float col = (wf[q][0][d] + wf[q][1][d]) + (wf[q][2][d] + wf[q][3][d]);
            acc[q][OUT][d] += (double)row;
            acc[q][OUT + 1][d] += (double)col;
            tot += row;
          }
          s1[q] = (double)tot;
        }
      } else
#pragma unroll
      for (int d2 = 0; d2 < 4; ++d2) {
        float p2[P];
        double s2[P];
#pragma unroll
        for (int q = 0; q < P; ++q) {
          p2[q] = po[q] * s_pwm[q][(G::PB + OUT) * 4 + d2];
          s2[q] = 0.0;
        }
#pragma unroll
        for (int d3 = 0; d3 < 4; ++d3) {
          float p3[P];
          double w[P];
#pragma unroll
          for (int q = 0; q < P; ++q) p3[q] = p2[q] * s_pwm[q][(G::PB + OUT + 1) * 4 + d3];
          const uint32_t x = xo | ((uint32_t)d2 << (2 * (G::PB + OUT))) | ((uint32_t)d3 << (2 * (G::PB + OUT + 1)));
          leaf(x, p3, w);
#pragma unroll
          for (int q = 0; q < P; ++q) {
            acc[q][OUT + 1][d3] += w[q];
            s2[q] += w[q];
          }
        }
#pragma unroll
        for (int q = 0; q < P; ++q) {
          acc[q][OUT][d2] += s2[q];
          s1[q] += s2[q];
        }
      }
      // the outer digits are run-time values: a select per cell keeps the accumulators in registers (indexing
      // acc[..][dj] would send the whole array to scratch memory).  Digit jo changes every 4^jo blocks: its cells
      // are updated then, from a running sum.
#pragma unroll
      for (int q = 0; q < P; ++q) {
#pragma unroll
        for (int jo = 0; jo < OUT; ++jo) {
          so[q][jo] += s1[q];
          if ((o & ((1 << (2 * jo)) - 1)) == (1 << (2 * jo)) - 1) {  // uniform; always true for jo = 0
            const int dj = (o >> (2 * jo)) & 3;
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[q][jo][a] += (a == dj) ? so[q][jo] : 0.0;
            so[q][jo] = 0.0;
          }
        }
        S[q] += s1[q];
      }
    }
  }

  // ---- block reduction (fixed tree) --------------------------------------------------------------
  // Digits 0..2 of the pattern id are lane bits, digit 3 is the wave index.  Cell (p, a) for p < 3 is the sum of
  // S over the lanes whose digit p equals a: an xor butterfly over the four lane bits outside digit p leaves it
  // in every lane of the class; lane a << 2p publishes it.  (A serial pass over 256 LDS values per cell was most
  // of the kernel for small PWM batches.)
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int q = 0; q < P; ++q) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      double v = S[q];
#pragma unroll
      for (int b = 0; b < 6; ++b)
        if ((b >> 1) != p) v += __shfl_xor(v, 1 << b, 64);
      if ((lane & ~(3 << (2 * p))) == 0) s_cls[q][wave][p][(lane >> (2 * p)) & 3] = v;
      if (p == 0) {  // whole wave
        double t = v;
        t += __shfl_xor(t, 1, 64);
        t += __shfl_xor(t, 2, 64);
        if (lane == 0) s_T[q][wave] = t;
      }
    }
    if constexpr (G::HI > 0) {
#pragma unroll
      for (int h = 0; h < G::HI; ++h)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const double v = wave_sum(acc[q][h][a]);
          if (lane == 0) s_red[q][wave][h * 4 + a] = v;
        }
    }
  }
  __syncthreads();
  if (tid < (uint32_t)G::CELLS) {
    const int p = tid >> 2, a = tid & 3;
#pragma unroll
    for (int q = 0; q < P; ++q) {
      if (!live[q]) continue;
      double v = 0.0;
      if (p < 3) {
        v = ((s_cls[q][0][p][a] + s_cls[q][1][p][a]) + s_cls[q][2][p][a]) + s_cls[q][3][p][a];
      } else if (p == 3) {
        v = s_T[q][a];
      } else if (p < G::PB) {  // digit fixed by the block index
        if ((int)((mid >> (2 * (p - G::LO))) & 3u) == a) v = ((s_T[q][0] + s_T[q][1]) + s_T[q][2]) + s_T[q][3];
      } else {
        if constexpr (G::HI > 0) {
          const int h = p - G::PB;
          v = ((s_red[q][0][h * 4 + a] + s_red[q][1][h * 4 + a]) + s_red[q][2][h * 4 + a]) + s_red[q][3][h * 4 + a];
        }
      }
      partials[((size_t)(pw0 + q) * G::NB + blockIdx.x) * G::CELLS + tid] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// em_fast = 2, "serial": the reference's float32 arithmetic INCLUDING its summation order, bit for bit.
// The reference adds the 4^W weights into each PWM cell one after the other in float32 (src/peng.cpp:121-127); the
// motifs' merge and redundancy decisions downstream compare similarity scores that are exactly tied in real arithmetic for
// reverse-complement twins, so the last bits of those sums decide what the program prints.  A cell's sum is inherently
// sequential as written, but its chain of roundings has structure a wave can use (seqsum.h).
//
// (the scheme's shared pieces -- span geometry, binade estimate, lean division, finalize step, LDS layout: em_serial.h)
// {min, max} of the background table as float bits (non-negative floats order like their bits; a negative entry or a NaN
// has the sign or all exponent bits set and ends up as a "max" no range test accepts).  Once per pengk_em call.
__global__ __launch_bounds__(1024) void em_bg_range_kernel(const float* __restrict__ bg, uint32_t np, uint32_t* __restrict__ range) {
  // (32 workgroups, one pair of atomics each: same-address device atomics queue up at ~20 ns apiece -- with one pair per
  // wave of a 256 x 256 grid this kernel took 29 us, six times the weights' saving per iteration)
  __shared__ uint32_t s_lo[16], s_hi[16];
  uint32_t lo = 0xFFFFFFFFu, hi = 0u;
  const uint4* v = reinterpret_cast<const uint4*>(bg);
  for (uint32_t i = blockIdx.x * 1024u + threadIdx.x; i < np / 4u; i += gridDim.x * 1024u) {
    const uint4 b = v[i];
    lo = min(min(lo, b.x), min(min(b.y, b.z), b.w));
    hi = max(max(hi, b.x), max(max(b.y, b.z), b.w));
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) {
    lo = min(lo, (uint32_t)__shfl_xor((int)lo, m, 64));
    hi = max(hi, (uint32_t)__shfl_xor((int)hi, m, 64));
  }
  if ((threadIdx.x & 63u) == 0u) {
    s_lo[threadIdx.x >> 6] = lo;
    s_hi[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) {
      lo = min(lo, s_lo[w]);
      hi = max(hi, s_hi[w]);
    }
    atomicMin(range, lo);
    atomicMax(range + 1, hi);
  }
}

// Self-test of lean_div (pengk_selftest_division): random operand pairs inside lean_div_ok's domain -- exponents over the
// whole range the guard admits, random mantissas, and the special mantissas (all zeros / all ones / one bit) where a
// division is likeliest to round the other way -- the unscaled sequence against the compiler's IEEE division.
__global__ __launch_bounds__(256) void em_div_check_kernel(unsigned long long seed, uint32_t per_thread, unsigned long long* __restrict__ out) {
  unsigned long long st = seed + 0x9E3779B97F4A7C15ull * (blockIdx.x * 256ull + threadIdx.x + 1ull);
  auto next = [&]() {
    unsigned long long z = (st += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  };
  unsigned long long checked = 0, bad = 0, zero = 0;
  for (uint32_t i = 0; i < per_thread; ++i) {
    const unsigned long long r = next(), q = next();
    uint32_t ma = (uint32_t)r & 0x7FFFFFu, mb = (uint32_t)(r >> 23) & 0x7FFFFFu;
    const uint32_t kind = (uint32_t)(q >> 60);
    if (kind == 0u) ma = 0u;
    else if (kind == 1u) mb = 0u;
    else if (kind == 2u) ma = 0x7FFFFFu;
    else if (kind == 3u) mb = 0x7FFFFFu;
    else if (kind == 4u) mb = 1u << ((q >> 40) % 23u);
    const uint32_t ea = 1u + (uint32_t)(q % 254ull), eb = 1u + (uint32_t)((q >> 8) % 254ull);
    float a = __uint_as_float((ea << 23) | ma);
    const float b = __uint_as_float((eb << 23) | mb);
    if (!lean_div_ok(a, a, b, b)) continue;
    if (((q >> 50) & 63ull) == 0ull) {  // a numerator of zero (a count of zero) is part of the domain
      a = 0.0f;
      ++zero;
    }
    const float lean = lean_div(a, b);
    float full;
    asm volatile("" : "+v"(a));  // (the two divisions are not to be merged)
    full = a / b;
    ++checked;
    bad += __float_as_uint(lean) != __float_as_uint(full);
  }
  atomicAdd(out, checked);
  atomicAdd(out + 1, bad);
  atomicAdd(out + 2, zero);
}

// The weights of a span, as em_weights_kernel computes them, and on the way the span's block sums.  A workgroup per span:
// thread t = digits 0..3 of x (lane = digits 0..2: a wave stores 64 consecutive floats), 64 x per thread over digits 4..6.
// The product over the PWM columns in the reference's order ((1*pwm[0][x0])*pwm[1][x1])... (src/peng.cpp:180-197): the
// factors of digits 0..3 once per thread, digit 4 once per 4 x, ...; the digits above the span, equal for all its x, are
// still multiplied last, x by x -- float products do not regroup.
// The kernel starts with the previous iteration's finalize step (fused_head: `fs`, launch `k`, `threshold`, `max_it`).
template <int W>
__global__ __launch_bounds__(256) void em_weights_span_kernel(const uint32_t* __restrict__ counts, const float* __restrict__ bg,
                                                              float saturation, float* __restrict__ wbuf,
                                                              float* __restrict__ sums, const uint32_t* __restrict__ bg_range,
                                                              FusedState fs, uint32_t k, float threshold, int max_it) {
  using G = BlockGeo<W>;
  PENGK_WG_TRACE_BEGIN(1);
  const uint32_t pw = blockIdx.y, sp = blockIdx.x;
  __shared__ __attribute__((aligned(16))) float s_pwm[W * 4], s_old[W * 4];
  __shared__ float part[4][28];
  __shared__ uint32_t s_lean;
  const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  const uint32_t bg_lo = bg_range[0], bg_hi = bg_range[1];
  if (!fused_head<W>(fs, pw, k, threshold, max_it, sp == 0u, s_pwm, s_old, t)) return;
  uint32_t* bad = fs.bad + (size_t)(k & 1u) * fs.bad_stride;
  if (t == 0) s_lean = lean_ranges_ok<W>(s_pwm, bg_lo, bg_hi, saturation) ? 1u : 0u;  // (workgroup-uniform: one PWM, one table)
  __syncthreads();
  const bool lean = s_lean != 0u;
  float* out = wbuf + (size_t)pw * G::NP + (size_t)sp * 16384u;
  const uint32_t* cnt = counts + (size_t)sp * 16384u;
  const float* bgs = bg + (size_t)sp * 16384u;
  float p3 = 1.0f;
#pragma unroll
  for (int p = 0; p < 4; ++p) p3 = p3 * s_pwm[p * 4 + ((t >> (2 * p)) & 3u)];
  float hi[W - 7];  // the span's own digits 7 .. W-1 (wave-uniform)
#pragma unroll
  for (int p = 7; p < W; ++p) hi[p - 7] = s_pwm[p * 4 + ((sp >> (2 * (p - 7))) & 3u)];
  float f4_[4], f5_[4], f6_[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    f4_[a] = s_pwm[16 + a];
    f5_[a] = s_pwm[20 + a];
    f6_[a] = s_pwm[24 + a];
  }
  float c4[4] = {0, 0, 0, 0}, c5[4] = {0, 0, 0, 0}, c6[4] = {0, 0, 0, 0};
  bool flagged = false;
  auto body = [&](auto lean_tag) {
    constexpr bool LEAN = decltype(lean_tag)::value;
#pragma unroll 1
    for (uint32_t d6 = 0; d6 < 4u; ++d6) {
#pragma unroll
      for (uint32_t d5 = 0; d5 < 4u; ++d5) {
        float s5 = 0.0f;
#pragma unroll
        for (uint32_t d4 = 0; d4 < 4u; ++d4) {
          const uint32_t xl = t + 256u * (d4 + 4u * d5 + 16u * d6);
          float pr = ((p3 * f4_[d4]) * f5_[d5]) * f6_[d6];
#pragma unroll
          for (int p = 7; p < W; ++p) pr = pr * hi[p - 7];
          const float odds = em_div<LEAN>(pr, bgs[xl]);
          const float v = em_div<LEAN>((float)cnt[xl] * saturation, 1 + em_div<LEAN>(saturation, odds));  // :124-125
          out[xl] = v;
          flagged |= __float_as_uint(v) > 0x7F7FFFFFu;
          c4[d4] += v;
          s5 += v;
        }
        c5[d5] += s5;
        c6[d6] += s5;
      }
    }
  };
  if (lean) body(std::true_type{});
  else body(std::false_type{});
  if (flagged) bad[pw] = 1u;  // (as em_weights_kernel: this PWM's cells are summed by the finalize kernel's plain loop)
  const float tot = (c6[0] + c6[1]) + (c6[2] + c6[3]);
  // per wave: whole-wave sums by digit 4, 5, 6; the total by digit 0, 1, 2 (lane bits 0-1, 2-3, 4-5); the total (digit 3)
  auto all = [](float v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
  };
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const float s4 = all(c4[a]), s5 = all(c5[a]), s6 = all(c6[a]);
    if (lane == 0) {
      part[wave][12 + a] = s4;
      part[wave][16 + a] = s5;
      part[wave][20 + a] = s6;
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {  // digit d = lane bits 2 d, 2 d + 1: add over the other four bits
    float v = tot;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1)
      if (m != (1 << (2 * d)) && m != (2 << (2 * d))) v += __shfl_xor(v, m, 64);
    if ((lane & ~(3u << (2 * d))) == 0u) part[wave][4 * d + (lane >> (2 * d))] = v;
  }
  {
    const float v = all(tot);
    if (lane == 0) part[wave][24] = v;
  }
  __syncthreads();
  if (t < G::CELLS) {
    const uint32_t p = t >> 2, a = t & 3u;
    float* o = sums + (size_t)pw * G::CELLS * G::NBLK;
    if (p <= 6u) {
      float v;
      if (p == 3u) v = part[a][24];  // digit 3 = the wave
      else {
        const uint32_t at = p <= 2u ? 4u * p + a : 4u * (p - 1u) + a;  // digits 4, 5, 6 at 12, 16, 20
        v = (part[0][at] + part[1][at]) + (part[2][at] + part[3][at]);
      }
      o[(size_t)t * G::NBLK + sp] = v;
    } else {
      // quarter a of the span: block high_block(p, sp, a) of cell (p, digit p of the span)
      const float v = (part[0][20 + a] + part[1][20 + a]) + (part[2][20 + a] + part[3][20 + a]);
      o[(size_t)(4u * p + G::high_digit(p, sp)) * G::NBLK + G::high_block(p, sp, a)] = v;
    }
  }
  PENGK_WG_TRACE_END(0, blockIdx.x + gridDim.x * blockIdx.y);
}

// The prefix of a cell's block sums -> block_binade of every block (cells of more than 1024 blocks; the shorter ones are
// predicted by em_span_eval_kernel itself).  One wave per cell; a lane takes NBLK / 64 consecutive blocks.
template <int W>
__global__ __launch_bounds__(64) void em_block_predict_kernel(const uint32_t* __restrict__ run_now, const uint32_t* __restrict__ bad,
                                                              const float* __restrict__ sums, seqsum::BlockRecord* __restrict__ rec,
                                                              uint32_t skew) {
  using G = BlockGeo<W>;
  const uint32_t pw = blockIdx.y, cell = blockIdx.x, lane = threadIdx.x;
  if (run_now[pw] == 0u || bad[pw]) return;
  constexpr uint32_t PER = G::NBLK / 64u;
  static_assert(G::NBLK % 64u == 0u, "whole lanes");
  const float* in = sums + ((size_t)pw * G::CELLS + cell) * G::NBLK + (size_t)lane * PER;
  seqsum::BlockRecord* out = rec + ((size_t)pw * G::CELLS + cell) * G::NBLK + (size_t)lane * PER;
  double mine = 0.0;
  for (uint32_t i = 0; i < PER; ++i) mine += (double)in[i];
  double before = mine;  // inclusive scan over the lanes, then exclusive
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double o = __shfl_up(before, d, 64);
    if ((int)lane >= d) before += o;
  }
  before -= mine;
  for (uint32_t i = 0; i < PER; ++i) {
    const double after = before + (double)in[i];
    const uint32_t e = block_binade(before, after, skew, cell * G::NBLK + lane * PER + i);
    seqsum::BlockRecord r;
    r.e = e;
    r.d0 = 0.0f;
    r.d1 = 0.0f;
    r.pad = 0u;
    out[i] = r;
    before = after;
  }
}

// One workgroup per span: the span's 64 KiB are read ONCE (not once per position) into LDS, and the eight waves share
// the 4 W blocks that lie in it -- each block with a predicted binade gets its two increments.
constexpr uint32_t SPAN_EVAL_WAVES = 8;
template <int W>
__global__ __launch_bounds__(64 * SPAN_EVAL_WAVES) void em_span_eval_kernel(const uint32_t* __restrict__ run_now, const float* __restrict__ wbuf,
                                                                            seqsum::BlockRecord* __restrict__ rec,
                                                                            const uint32_t* __restrict__ bad, uint32_t n_pwm,
                                                                            const float* __restrict__ sums, uint32_t skew,
                                                                            uint32_t extra_wgs, uint32_t head_blocks) {
  using G = BlockGeo<W>;
  // (run_now: the PWMs' "still running" flags behind this iteration's head, FusedState::run)
  // (consecutive workgroups go to consecutive XCDs: a PWM's spans, and behind them its chains, stay on one -- as in
  // em_fold_scan_kernel; 1000 PWMs x 10 iterations: 35.5 ms, with PWM = blockIdx.y 37.5)
  PENGK_WG_TRACE_BEGIN(2);
  const uint32_t lin = blockIdx.x + gridDim.x * blockIdx.y;
  __shared__ __attribute__((aligned(16))) float span[16384];
  const uint32_t t = threadIdx.x, lane = t & 63u;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(t >> 6));
  if (lin < extra_wgs) {
    // The workgroups IN FRONT of the spans' (they start first; behind them they were the kernel's tail: 29 -> 36 us) fold
    // BLOCK 0 of every cell: its start is known exactly (zero), and it is the
    // dearest block of a chain -- the sum climbs through some twenty binades in it, each crossing another evaluation --
    // so it is folded here, beside the evaluation of all the other blocks, instead of at the head of every chain
    // (the chains: 37 -> 30 us per iteration for 16 PWMs).  Four waves per workgroup (a block staged per wave in a
    // quarter of the span buffer), a cell each; the record says SUM_BEHIND and carries the sum.
    constexpr uint32_t XW = (G::CELLS + 3u) / 4u;  // workgroups per PWM
    uint32_t pw, xunit;
    if (!group_map(lin, n_pwm, XW, pw, xunit)) return;
    const uint32_t cell = 4u * xunit + wave;
    if (wave >= 4u || cell >= G::CELLS || run_now[pw] == 0u || bad[pw]) return;
    seqsum::lds_float* buf = (seqsum::lds_float*)span + wave * seqsum::BLOCK;
    const float* w = wbuf + (size_t)pw * G::NP;
    // ... and the `head_blocks` - 1 blocks behind it, one after the other from the exact sum: the sum doubles from block
    // to block there (1 -> 2 -> 4 blocks' worth), so these are the blocks a chain would take the long way, several
    // evaluations each; here they cost nothing on the iteration's critical path as long as these workgroups end before the
    // spans' do.  A block is asked for as soon as the one before it has left the buffer for the registers.
    // (ONE place that holds a Row and calls fold_block: a second instantiation costs ~30 registers and with them the
    // fourth wave per SIMD of every workgroup of this kernel -- the spans' included: 16 PWMs 0.77 -> 0.83 ms, W = 12 10.0 -> 12.7)
    seqsum::Row mine;
    seqsum::Stats st;
    float s0 = 0.0f;
    const bool first_position = (cell >> 2) == 0u;  // (wave-uniform)
    auto stage = [&](uint32_t b) {
      // (the sources' per-lane offsets are worked out again for every block -- from a lane number the compiler cannot
      // see through -- so that they are not sixteen more registers alive across fold_block)
      uint32_t l = lane;
      asm volatile("" : "+v"(l));
      if (first_position) {
        EmTerms0<W> src0{w, cell & 3u};
        src0.bind_stage(l);
        src0.stage(b, l, buf);
      } else {
        EmTerms<W> src{w, cell >> 2, cell & 3u};
        src.bind_stage(l);
        src.stage(b, l, buf);
      }
    };
    stage(0u);
#pragma unroll 1
    for (uint32_t b = 0; b < head_blocks; ++b) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      mine.read_staged(buf, lane);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (b + 1u < head_blocks) stage(b + 1u);
      s0 = seqsum::fold_block(mine, lane, s0, st);
    }
    if (lane == 0) {
      seqsum::BlockRecord out;
      out.e = seqsum::SUM_BEHIND;
      out.d0 = s0;
      out.d1 = 0.0f;
      out.pad = head_blocks;  // (the chain starts behind them: seqsum::walk_chain)
      rec[((size_t)pw * G::CELLS + cell) * G::NBLK] = out;
    }
    PENGK_WG_TRACE_END(1, lin);
    return;
  }
  uint32_t pw, sp;
  if (!group_map(lin - extra_wgs, n_pwm, G::SPANS, pw, sp)) return;
  // The span's 64 KiB are asked for FIRST, together with the PWM's flags; the estimates below (another round trip: the
  // cells' block sums) are worked out while the span is on its way.
  constexpr uint32_t T = 64u * SPAN_EVAL_WAVES, PER = 4096u / T;
  seqsum::f4 v[PER];
  {
    const seqsum::f4* src = reinterpret_cast<const seqsum::f4*>(wbuf + (size_t)pw * G::NP + (size_t)sp * 16384u);
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) v[k] = src[t + T * k];
  }
  // (the flags are looked at behind the estimates -- which touch nothing but this workgroup's LDS -- so that no branch
  // stands between the span's loads and the estimates' loads: a workgroup of a PWM that is done leaves a little later)
  const uint32_t running = run_now[pw];
  const uint32_t flagged = bad[pw];
  constexpr uint32_t TASKS = (G::CELLS + SPAN_EVAL_WAVES - 1u) / SPAN_EVAL_WAVES;  // per wave
  seqsum::BlockRecord* cells = rec + (size_t)pw * G::CELLS * G::NBLK;
  // task (p, j) of the span -> its cell and block
  auto cell_of = [&](uint32_t task) { return (task >> 2) <= 6u ? task : 4u * (task >> 2) + G::high_digit(task >> 2, sp); };
  auto block_of = [&](uint32_t task) { return (task >> 2) <= 6u ? sp : G::high_block(task >> 2, sp, task & 3u); };
  // the binades of this wave's blocks: the estimate in front of / behind a block from the cell's block sums
  // (block_binade), or what em_block_predict_kernel left in the record
  // (kept in a register, lane i = this wave's i-th block: with nothing in LDS beside the span a workgroup takes exactly
  // 64 KiB, and two of them fit a CU's 160 KiB beside a chain workgroup's 32 KiB of the other lane -- 1 % on every EM figure)
  uint32_t binades = seqsum::NO_BINADE;
  if constexpr (G::PREDICT_IN_EVAL) {
    // The block sums of SEVERAL of this wave's cells are asked for together, then reduced: one task after the other -- load,
    // wait, reduce, next -- the five round trips stood in front of every span's evaluation (5.8 of a workgroup's 10.9 us,
    // tools/em_wgtrace.py).  All tasks at once where a cell has up to 256 blocks, two at a time above that (16 loads each).
    constexpr uint32_t LOADS = G::NBLK / 64u, GROUP = LOADS <= 4u ? TASKS : 2u;
#pragma unroll 1
    for (uint32_t i0 = 0; i0 < TASKS; i0 += GROUP) {
      float part[GROUP], own[GROUP];
#pragma unroll
      for (uint32_t g = 0; g < GROUP; ++g) {
        const uint32_t task = min(wave + SPAN_EVAL_WAVES * (i0 + g), G::CELLS - 1u);  // (a task past the last cell: looked at by nobody)
        const uint32_t cell = cell_of(task), b = block_of(task);
        const float* cs = sums + ((size_t)pw * G::CELLS + cell) * G::NBLK;
        float acc = 0.0f;
#pragma unroll
        for (uint32_t c = 0; c < G::NBLK; c += 64u) {
          const float x = cs[c + lane];
          acc += c + lane < b ? x : 0.0f;
        }
        part[g] = acc;
        own[g] = cs[b];
      }
#pragma unroll
      for (uint32_t g = 0; g < GROUP; ++g) {
        const uint32_t i = i0 + g, task = wave + SPAN_EVAL_WAVES * i;
        float acc = part[g];
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) acc += __shfl_xor(acc, m, 64);
        if (i < TASKS && task < G::CELLS) {
          const uint32_t e = block_binade((double)acc, (double)acc + (double)own[g], skew, cell_of(task) * G::NBLK + block_of(task));  // (the same in all lanes)
          if (lane == i) binades = e;
        }
      }
    }
  } else {
#pragma unroll 1
    for (uint32_t i = 0; i < TASKS; ++i) {
      const uint32_t task = wave + SPAN_EVAL_WAVES * i;
      if (task < G::CELLS) {
        const uint32_t e = cells[(size_t)cell_of(task) * G::NBLK + block_of(task)].e;
        if (lane == i) binades = e;
      }
    }
  }
  if (running == 0 || flagged) return;
  {
    seqsum::f4* dst = reinterpret_cast<seqsum::f4*>(span);
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
      const uint32_t idx = t + T * k;
      dst[SpanLds::slot_of(idx >> 4, idx & 15u)] = v[k];
    }
  }
  __syncthreads();
  PENGK_WG_TRACE_END(2, lin);  // (the span is in LDS)
#pragma unroll 1
  for (uint32_t task = wave, i = 0; task < G::CELLS; task += SPAN_EVAL_WAVES, ++i) {
    const uint32_t p = task >> 2, j = task & 3u;
    if (block_of(task) < head_blocks) continue;  // (folded from zero by the workgroups in front of the spans')
    seqsum::BlockRecord* r = cells + (size_t)cell_of(task) * G::NBLK + block_of(task);
    const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)binades, (int)i);
    if (e == seqsum::NO_BINADE) {
      if (G::PREDICT_IN_EVAL && lane == 0) r->e = seqsum::NO_BINADE;
      continue;
    }
    seqsum::Row mine;
    span_row<W>(span, p, j, lane, mine);
    float d0, d1;
    const bool ok = seqsum::block_increments(mine, lane, seqsum::bases_of_binade(e), d0, d1);
    if (lane == 0) {
      seqsum::BlockRecord out;
      out.e = ok ? e : seqsum::NO_BINADE;
      out.d0 = d0;
      out.d1 = d1;
      out.pad = 0u;
      *r = out;
    }
  }
  PENGK_WG_TRACE_END(0, lin);
}

// The chains of an iteration: one wave per cell; the cell's sum is stored for the next iteration's head
// (or em_serial_finish_kernel) -- a plain store, the kernel boundary orders it.
template <int W>
__global__ __launch_bounds__(64) void em_chain_store_kernel(const uint32_t* __restrict__ run, const uint32_t* __restrict__ bad,
                                                            const float* __restrict__ wbuf, const seqsum::BlockRecord* __restrict__ rec,
                                                            float* __restrict__ cellsum, uint32_t n_pwm,
                                                            unsigned long long* __restrict__ counters) {
  using G = BlockGeo<W>;
  PENGK_WG_TRACE_BEGIN(3);
  const uint32_t lin = blockIdx.x + gridDim.x * blockIdx.y;
  uint32_t pw, cell;
  if (!group_map(lin, n_pwm, G::CELLS, pw, cell)) return;
  __shared__ __attribute__((aligned(16))) float lds[seqsum::WALK_LDS_FLOATS];
  seqsum::WalkCounts wc;
  const uint32_t lane = threadIdx.x;
  const seqsum::BlockRecord* r = rec + ((size_t)pw * G::CELLS + cell) * G::NBLK;
  // (the PWM's two flags and the chain's first 64 records are asked for together: one memory round trip at the head of
  // every chain -- the kernel ends with its longest one)
  uint32_t running = run[pw];
  uint32_t flagged = bad[pw];
  uint4 first = reinterpret_cast<const uint4*>(r)[lane];
  asm volatile("" : "+s"(running), "+s"(flagged), "+v"(first.x), "+v"(first.y), "+v"(first.z), "+v"(first.w));
  if (running == 0u) return;
  float s = 0.0f;
  const float* w = wbuf + (size_t)pw * G::NP;
  if (flagged) {
    // a PWM with a negative or non-finite weight (degenerate inputs only): the plain loop, the reference's own additions
    const uint32_t p = cell >> 2, a = cell & 3u;
    for (uint32_t c = 0; c < (1u << (2 * W - 2)); ++c)
      s += w[((c >> (2u * p)) << (2u * p + 2u)) | (a << (2u * p)) | (c & ((1u << (2u * p)) - 1u))];
  } else if ((cell >> 2) == 0u) {
    EmTerms0<W> src0{w, cell & 3u};
    src0.bind_stage(lane);
    s = seqsum::walk_chain(src0, r, first, G::NBLK, (seqsum::lds_float*)lds, lane, wc);
  } else {
    EmTerms<W> src{w, cell >> 2, cell & 3u};
    src.bind_stage(lane);
    s = seqsum::walk_chain(src, r, first, G::NBLK, (seqsum::lds_float*)lds, lane, wc);
  }
  if (lane == 0) {
    cellsum[(size_t)pw * G::CELLS + cell] = s;
    if (wc.fetched) __hip_atomic_fetch_add(counters + 0, (unsigned long long)wc.fetched, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wc.mispredicted) __hip_atomic_fetch_add(counters + 1, (unsigned long long)wc.mispredicted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wc.restaged) __hip_atomic_fetch_add(counters + 2, (unsigned long long)wc.restaged, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wc.restaged_waits) __hip_atomic_fetch_add(counters + 3, (unsigned long long)wc.restaged_waits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  PENGK_WG_TRACE_END(wc.fetched > 255u ? 255u : wc.fetched, lin);
}

// Once per call and batch, behind the last launch's chains: F_K for the PWMs that are still running (K = launches = the
// iteration limit: they stop here), and every PWM's final matrix into the caller's array (PWM_j lives there for even j).
template <int W>
__global__ __launch_bounds__(64) void em_serial_finish_kernel(FusedState fs, uint32_t K, float threshold, int max_it) {
  constexpr uint32_t CELLS = 4u * W;
  const uint32_t pw = blockIdx.x, t = threadIdx.x;
  __shared__ __attribute__((aligned(16))) float s_pwm[CELLS], s_old[CELLS];
  // run[(K-1) & 1] = running behind F_(K-1) -- what the last launch (k = K) wrote, or the initial flag at K = 1
  const uint32_t running = fs.run[(size_t)((K - 1u) & 1u) * fs.run_stride + pw];
  if (running) {
    const float* prev = ((K - 1u) & 1u) ? fs.pwm1 : fs.pwm0;  // PWM_(K-1)
    if (t < CELLS) {
      s_old[t] = prev[(size_t)pw * CELLS + t];
      s_pwm[t] = fs.cellsum[(size_t)pw * CELLS + t];
    }
    __syncthreads();
    const float change = finalize_rows<W>(s_pwm, s_old, t);
    if (t < CELLS) fs.pwm0[(size_t)pw * CELLS + t] = s_pwm[t];
    if (t == 0) {
      fs.state[2 * pw] = (int)K;
      fs.state[2 * pw + 1] = !(change <= threshold || (int)K >= max_it);
      fs.change[pw] = change;
    }
  } else {
    const int it = fs.state[2 * pw];  // the PWM stopped behind F_it: PWM_it
    if ((it & 1) && t < CELLS) fs.pwm0[(size_t)pw * CELLS + t] = fs.pwm1[(size_t)pw * CELLS + t];
  }
  if (t == 0) {
    fs.bad[pw] = 0u;
    fs.bad[(size_t)fs.bad_stride + pw] = 0u;
  }
}

// What a call in this mode starts from: the caller's state as em_init_kernel leaves it, both copies of the "running"
// flag, the "bad weight" flags, the chains' counters, the background's range for em_bg_range_kernel's min / max.
__global__ __launch_bounds__(256) void em_serial_setup_kernel(uint32_t n, int W, float threshold, int max_it, int32_t* __restrict__ state,
                                                             float* __restrict__ change, uint32_t* __restrict__ run,
                                                             unsigned long long* __restrict__ counters, uint32_t* __restrict__ bg_range,
                                                             uint32_t lean, uint32_t* __restrict__ bad, uint32_t bad_words) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < (uint32_t)EM_COUNTERS) counters[i] = 0ull;
  if (i == 0) {
    bg_range[0] = lean ? 0xFFFFFFFFu : 0u;
    bg_range[1] = lean ? 0u : 0xFFFFFFFFu;
  }
  for (uint32_t j = i; j < bad_words; j += gridDim.x * 256u) bad[j] = 0u;
  if (i >= n) return;
  const float c0 = (float)W;  // `float change = pattern_length` (src/peng.cpp:101)
  const uint32_t r = !(c0 <= threshold || 0 >= max_it);
  state[2 * i] = 0;
  state[2 * i + 1] = (int32_t)r;
  change[i] = c0;
  run[i] = r;
  run[n + i] = r;
}

template <int W, int HIMAX, bool FAST, int P>
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
  if (batch > P) batch -= batch % P;   // whole groups of P PWMs per batch
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
      hipLaunchKernelGGL((em_accumulate_kernel<W, HIMAX, FAST, P>), dim3(G::NB, (unsigned)((nb + P - 1) / P)), dim3(256), 0,
                         ctx->stream, d_pwms + (size_t)first * W * 4, d_state + 2 * first, d_counts, d_bg, saturation,
                         ctx->d_em_partials, (int)nb);
      hipLaunchKernelGGL((em_finalize_kernel<W, HIMAX>), dim3((unsigned)nb), dim3(64), 0, ctx->stream,
                         d_pwms + (size_t)first * W * 4, d_state + 2 * first, d_change + first, ctx->d_em_partials, threshold, max_it,
                         (uint32_t*)nullptr, (const float*)nullptr, 0u);
    }
    PENGK_HIP(hipGetLastError());
  }
  return PENGK_OK;
}

#ifdef PENGK_WG_TRACE
}  // namespace
}  // namespace pengk
// copies up to `max` records (3 words each) to `out`, returns how many there were, and starts over
extern "C" __attribute__((visibility("default"))) long long pengk_debug_wg_trace(unsigned long long* out, unsigned max) {
  unsigned n = 0;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(pengk::g_wg_trace_n), sizeof n) != hipSuccess) return -1;
  const unsigned m = n < max ? n : max;
  if (m && hipMemcpyFromSymbol(out, HIP_SYMBOL(pengk::g_wg_trace), (size_t)m * 3 * sizeof(unsigned long long)) != hipSuccess) return -1;
  const unsigned zero = 0;
  if (hipMemcpyToSymbol(HIP_SYMBOL(pengk::g_wg_trace_n), &zero, sizeof zero) != hipSuccess) return -1;
  return (long long)n;
}
namespace pengk {
namespace {
#endif

#ifdef PENGK_SEQSUM_STATS
}  // namespace
}  // namespace pengk
extern "C" __attribute__((visibility("default"))) int pengk_debug_seqsum_stats(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pengk::seqsum::g_stats), 12 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[12] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(pengk::seqsum::g_stats), z, sizeof z) != hipSuccess) return 1;
  }
  return 0;
}
namespace pengk {
namespace {
#endif

// The serial mode with the blocks evaluated ahead of their chain (W >= 10), on SEVERAL streams: the PWMs go round in
// batches, and the batches take turns on the context's stream and up to three more ("lanes", option em_overlap).  A
// batch's iteration is weights -> block evaluation -> chains (`split`), or span kernel -> chains (em_serial_scan = 3); the
// first kernels are bound by arithmetic and the last by one wave per cell waiting for its next block: with several
// batches in flight the waits of one are filled by the others (16 PWMs x 10 iterations at W = 10: 0.92 / 0.83 / 0.86 / 0.84 ms
// on 1 / 2 / 3 / 4 streams in round 4; two by default).  The launches are ENQUEUED in turn as well -- iteration 1 of every
// lane's batch, then iteration 2 ... (batch by batch, the second lane got its first kernel when the host had enqueued the
// first lane's thirty launches: profiles/r04_em_kernels.log).  One em_serial_finish_kernel per batch.
template <int W>
int launch_serial_ahead(pengk_ctx* ctx, int64_t n_pwm, float* d_pwms, float saturation, float threshold, int max_it,
                        const uint32_t* d_counts, const float* d_bg, int32_t* d_state, float* d_change, size_t budget, bool split) {
  using B = BlockGeo<W>;
  using LG = LookGeo<W>;
  const size_t np = (size_t)1 << (2 * W);
  int lanes = ctx->em_overlap < 1 ? 1 : ctx->em_overlap > MAX_EM_LANES ? MAX_EM_LANES : ctx->em_overlap;
  while (lanes > 1 && n_pwm < 8 * (int64_t)lanes) --lanes;
  const int64_t fit = std::max<int64_t>(1, (int64_t)(budget / lanes / (np * sizeof(float))));  // tables the budget holds per lane
  int64_t batch = fit;
  if (batch * lanes > n_pwm) {
    batch = lanes > 1 ? ((n_pwm + lanes - 1) / lanes + 7) / 8 * 8 : n_pwm;
    if (batch > fit) batch = fit >= 8 ? fit / 8 * 8 : fit;
  }
  if (batch > 16384) batch = 16384;  // (grid size; far beyond any budget)
  // per lane: tables | records | look-back words | cell sums, "bad weight" flags.  Per call: run[2][n] | PWM_odd[n][4 W]
  const size_t tables_b = (size_t)batch * np * sizeof(float);
  const size_t sums_b = split ? ((size_t)batch * B::CELLS * B::NBLK * sizeof(float) + 255) / 256 * 256 : 0;  // (split: the plain block sums in front of the records)
  const size_t rec_b = sums_b + ((size_t)batch * B::CELLS * B::NBLK * sizeof(seqsum::BlockRecord) + 255) / 256 * 256;
  const size_t look_b = split ? 0 : ((size_t)batch * LG::WORDS_PER_PWM * sizeof(unsigned long long) + 255) / 256 * 256;
  const size_t flags_at = ((size_t)batch * B::CELLS * sizeof(float) + 255) / 256 * 256;
  const size_t small_b = (flags_at + (size_t)2 * batch * sizeof(uint32_t) + 255) / 256 * 256;
  const size_t run_at = lanes * small_b;
  const size_t pwm1_at = (run_at + (size_t)2 * n_pwm * sizeof(uint32_t) + 255) / 256 * 256;
  int rc = ensure_scratch(ctx, (void**)&ctx->d_em_tables, &ctx->em_tables_bytes, lanes * tables_b);
  if (rc) return rc;
  rc = ensure_scratch(ctx, (void**)&ctx->d_em_partials, &ctx->em_partials_bytes, pwm1_at + (size_t)n_pwm * B::CELLS * sizeof(float));
  if (rc) return rc;
  rc = ensure_scratch(ctx, &ctx->d_em_blocks, &ctx->em_blocks_bytes, lanes * rec_b);
  if (rc) return rc;
  if (look_b) {
    // the look-back words carry the epoch of the launch that wrote them: a fresh buffer starts from zero (no launch has
    // epoch 0), and so does a counter that has gone round
    const size_t had = ctx->em_look_bytes;
    rc = ensure_scratch(ctx, &ctx->d_em_look, &ctx->em_look_bytes, lanes * look_b);
    if (rc) return rc;
    if (ctx->em_look_bytes != had || ctx->em_epoch >= 0xFFFF0000u) {
      PENGK_HIP(hipStreamSynchronize(ctx->stream));
      for (int l = 0; l < MAX_EM_LANES - 1; ++l)
        if (ctx->em_streams[l]) PENGK_HIP(hipStreamSynchronize(ctx->em_streams[l]));
      PENGK_HIP(hipMemsetAsync(ctx->d_em_look, 0, ctx->em_look_bytes, ctx->stream));
      ctx->em_epoch = 0;
    }
  }
  if (!ctx->d_em_counters) PENGK_HIP(hipMalloc((void**)&ctx->d_em_counters, (EM_COUNTERS + 1) * sizeof(unsigned long long)));
  uint32_t* bg_range = reinterpret_cast<uint32_t*>(ctx->d_em_counters + EM_COUNTERS);
  char* small = reinterpret_cast<char*>(ctx->d_em_partials);
  uint32_t* run = reinterpret_cast<uint32_t*>(small + run_at);
  float* pwm1 = reinterpret_cast<float*>(small + pwm1_at);
  // (the lanes' flag words lie small_b apart: the setup kernel clears everything from the first lane's flags to the last lane's)
  {
    const uint32_t words = (uint32_t)((run_at - flags_at) / sizeof(uint32_t));
    hipLaunchKernelGGL(em_serial_setup_kernel, dim3((unsigned)((std::max<int64_t>(n_pwm, 1024) + 255) / 256)), dim3(256), 0, ctx->stream,
                       (uint32_t)n_pwm, W, threshold, max_it, d_state, d_change, run, ctx->d_em_counters, bg_range,
                       ctx->em_lean_div ? 1u : 0u, reinterpret_cast<uint32_t*>(small + flags_at), words);
    if (ctx->em_lean_div)
      hipLaunchKernelGGL(em_bg_range_kernel, dim3(32), dim3(1024), 0, ctx->stream, d_bg, (uint32_t)np, bg_range);  // (np = 4^W: a multiple of 4)
    PENGK_HIP(hipGetLastError());
  }
  if (max_it <= 0) return PENGK_OK;
  hipStream_t streams[MAX_EM_LANES];
  streams[0] = ctx->stream;
  for (int l = 1; l < lanes; ++l) {
    if (!ctx->em_streams[l - 1]) PENGK_HIP(hipStreamCreateWithFlags(&ctx->em_streams[l - 1], hipStreamNonBlocking));
    if (!ctx->em_join[l - 1]) PENGK_HIP(hipEventCreateWithFlags(&ctx->em_join[l - 1], hipEventDisableTiming));
    streams[l] = ctx->em_streams[l - 1];
  }
  if (lanes > 1 && !ctx->em_fork) PENGK_HIP(hipEventCreateWithFlags(&ctx->em_fork, hipEventDisableTiming));
  if (lanes > 1) {  // (everything enqueued so far -- the tables' producers, the setup -- comes first on all of them)
    PENGK_HIP(hipEventRecord(ctx->em_fork, ctx->stream));
    for (int l = 1; l < lanes; ++l) PENGK_HIP(hipStreamWaitEvent(streams[l], ctx->em_fork, 0));
  }
  // Once the lanes are forked they are ALWAYS joined, also when a launch fails half way (launch_serial_ahead).
  const int rc_launch = [&]() -> int {
    for (int64_t round0 = 0; round0 < n_pwm; round0 += batch * lanes) {
      for (int it = 1; it <= max_it + 1; ++it) {  // (turn max_it + 1: the batches' finish kernels)
        for (int l = 0; l < lanes; ++l) {
          const int64_t first = round0 + (int64_t)l * batch;
          if (first >= n_pwm) break;
          const int64_t nb = n_pwm - first < batch ? n_pwm - first : batch;
          hipStream_t st = streams[l];
          float* tables = reinterpret_cast<float*>(reinterpret_cast<char*>(ctx->d_em_tables) + l * tables_b);
          float* sums = reinterpret_cast<float*>(reinterpret_cast<char*>(ctx->d_em_blocks) + l * rec_b);
          seqsum::BlockRecord* rec = reinterpret_cast<seqsum::BlockRecord*>(reinterpret_cast<char*>(ctx->d_em_blocks) + l * rec_b + sums_b);
          unsigned long long* look = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(ctx->d_em_look) + l * look_b);
          float* cellsum = reinterpret_cast<float*>(small + l * small_b);
          uint32_t* bad = reinterpret_cast<uint32_t*>(small + l * small_b + flags_at);
          FusedState fs;
          fs.run = run + first;
          fs.pwm0 = d_pwms + (size_t)first * W * 4;
          fs.pwm1 = pwm1 + (size_t)first * W * 4;
          fs.bad = bad;
          fs.cellsum = cellsum;
          fs.state = d_state + 2 * first;
          fs.change = d_change + first;
          fs.n = (uint32_t)nb;
          fs.run_stride = (uint32_t)n_pwm;
          fs.bad_stride = (uint32_t)batch;
          if (it > max_it) {
            hipLaunchKernelGGL((em_serial_finish_kernel<W>), dim3((unsigned)nb), dim3(64), 0, st, fs, (uint32_t)max_it, threshold, max_it);
            continue;
          }
          const uint32_t k = (uint32_t)it;
          const unsigned groups = (unsigned)((nb + 7) / 8 * 8);  // (PWMs in whole groups of 8, one per XCD)
          const uint32_t* run_now = run + (size_t)((k - 1u) & 1u) * n_pwm + first;  // behind this launch's head
          const uint32_t* bad_now = bad + (size_t)(k & 1u) * batch;
          if (split) {
            hipLaunchKernelGGL((em_weights_span_kernel<W>), dim3(B::SPANS, (unsigned)nb), dim3(256), 0, st, d_counts, d_bg, saturation, tables,
                               sums, (const uint32_t*)bg_range, fs, k, threshold, max_it);
            if (!B::PREDICT_IN_EVAL)
              hipLaunchKernelGGL((em_block_predict_kernel<W>), dim3(B::CELLS, (unsigned)nb), dim3(64), 0, st, run_now, bad_now,
                                 (const float*)sums, rec, (uint32_t)ctx->em_test_skew);
            const uint64_t xwgs = (uint64_t)nb * ((B::CELLS + 3) / 4), swgs = xwgs + (uint64_t)nb * B::SPANS;  // (group_map: exactly the PWMs' workgroups)
            hipLaunchKernelGGL((em_span_eval_kernel<W>), dim3(1024u, (unsigned)((swgs + 1023u) / 1024u)), dim3(64 * SPAN_EVAL_WAVES), 0, st,
                               run_now, (const float*)tables, rec, bad_now, (uint32_t)nb, (const float*)sums,
                               (uint32_t)ctx->em_test_skew, (uint32_t)xwgs, (uint32_t)std::min<uint64_t>(ctx->em_head_blocks, B::NBLK));
            hipLaunchKernelGGL((em_chain_store_kernel<W>), dim3((unsigned)(4 * W), (unsigned)nb), dim3(64), 0, st, run_now, bad_now,
                               (const float*)tables, (const seqsum::BlockRecord*)rec, cellsum, (uint32_t)nb, ctx->d_em_counters);
            continue;
          }
          if constexpr (LG::SUPPORTED) {
            const uint64_t wgs = (uint64_t)groups * B::SPANS;
            const unsigned gx = 1024u;
            const uint32_t epoch = ++ctx->em_epoch;
            const int rc_fused = launch_span_fused(W, gx, (unsigned)((wgs + gx - 1) / gx), st, fs, k, threshold, max_it, d_counts, d_bg, saturation,
                                                   tables, rec, look, epoch, (const uint32_t*)bg_range, (uint32_t)ctx->em_test_skew,
                                                   (uint32_t)ctx->em_test_lookback);  // (em_fused.hip)
            if (rc_fused) return rc_fused;
          }
          hipLaunchKernelGGL((em_chain_store_kernel<W>), dim3((unsigned)(4 * W), (unsigned)nb), dim3(64), 0, st, run_now, bad_now,
                             (const float*)tables, (const seqsum::BlockRecord*)rec, cellsum, (uint32_t)nb, ctx->d_em_counters);
        }
      }
      PENGK_HIP(hipGetLastError());
    }
    return PENGK_OK;
  }();
  int rc_join = PENGK_OK;
  for (int l = 1; l < lanes; ++l) {
    hipError_t e = hipEventRecord(ctx->em_join[l - 1], streams[l]);
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ctx->em_join[l - 1], 0);
    if (e != hipSuccess) {
      (void)hipStreamSynchronize(streams[l]);  // the join could not be enqueued: wait here instead
      if (!rc_join) rc_join = hip_fail(e, "joining the EM's streams");
    }
  }
  return rc_launch ? rc_launch : rc_join;
}

#ifndef PENGK_EM_BUDGET_GIB
#define PENGK_EM_BUDGET_GIB 24
#endif
// The serial mode (em_fast = 2).  W >= 10: the blocks-ahead scheme of this file; W = 4 .. 8, and any W under the test hook
// pengk_test_em_generation (ctx->em_serial_scan = 0 / 1), the earlier generations of em_legacy.hip.
template <int W>
int launch_serial(pengk_ctx* ctx, int64_t n_pwm, float* d_pwms, float saturation, float threshold, int max_it,
                  const uint32_t* d_counts, const float* d_bg, int32_t* d_state, float* d_change) {
  const size_t np = (size_t)1 << (2 * W);
  // Weight tables of one batch of PWMs (4^W floats per PWM, twice that with the legacy scan's permuted copy: 8 MiB at W = 10,
  // 128 MiB at W = 12); a batch is one launch per kernel and iteration.  Option "em_table_budget_mb" (0 = automatic):
  //  * tables of up to 16 MiB per PWM (W <= 10): 192 MiB per batch, so that what the weights kernel writes is still in
  //    the 256 MiB Infinity Cache when it is read again (1000 PWMs x 10 iterations at W = 10 with the round-3 scan: 48.6 ms;
  //    with 1 GiB batches 58 ms, with all PWMs in one batch 79 ms; profiles/r03_em_budget.log);
  //  * larger tables never fit: one batch for the whole call (a quarter of the free memory at most), i.e. one launch
  //    per kernel and iteration.
  size_t budget = (size_t)ctx->em_table_budget_mb << 20;
  if (budget == 0) {
    if (2 * np * sizeof(float) <= ((size_t)16 << 20)) {
      budget = (size_t)192 << 20;
    } else {
      size_t free_b = 0, total_b = 0;
      budget = (size_t)1 << 30;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
        budget = std::min(std::max(budget, (free_b + ctx->em_tables_bytes) / 4), (size_t)PENGK_EM_BUDGET_GIB << 30);
    }
  }
  if constexpr (W >= 10) {
    // 2 (the library's scheme): weights (with the previous iteration's finalize step at their head), block evaluation,
    // chains; 3: weights and block evaluation as one kernel -- up to W = 12: 16384 spans per PWM would take a third look-back level
    if (ctx->em_serial_scan >= 2)
      return launch_serial_ahead<W>(ctx, n_pwm, d_pwms, saturation, threshold, max_it, d_counts, d_bg, d_state, d_change, budget,
                                    ctx->em_serial_scan != 3 || !LookGeo<W>::SUPPORTED);
  }
  return launch_em_serial_legacy(ctx, W, ctx->em_serial_scan != 0 ? 1 : 0, n_pwm, d_pwms, saturation, threshold, max_it, d_counts, d_bg,
                                 d_state, d_change, budget);
}

// W = 2 (16 patterns, 8 cells of 4 weights): the whole EM of a PWM in one wave -- the geometry above takes four digits of
// the pattern from the thread index.  Lane x < 16 owns pattern x; per iteration: its weight with the reference's float
// operations (src/peng.cpp:124-125, 180-197; mode 1: the throughput mode's one-reciprocal form), the cells' sums -- mode 2:
// float32 in ascending x, the reference's order (:121-127); modes 0 and 1: fp64 -- and the reference's float32 epilogue
// (row normalisation :129, change :132-137, swap :140-143) on lane 0, until the PWM has converged (:104).
__global__ __launch_bounds__(64) void em_w2_kernel(float* __restrict__ pwms, int32_t* __restrict__ state, float* __restrict__ change_out,
                                                   const uint32_t* __restrict__ counts, const float* __restrict__ bg, float saturation,
                                                   float threshold, int max_it, int mode) {
  const int pw = blockIdx.x, lane = threadIdx.x;
  __shared__ float s_pwm[8], s_w[16];
  __shared__ int s_active;
  float* old = pwms + (size_t)pw * 8;
  if (lane < 8) s_pwm[lane] = old[lane];
  if (lane == 0) s_active = state[2 * pw + 1];
  __syncthreads();
  const float cs = lane < 16 ? (float)counts[lane] * saturation : 0.0f;
  const float b = lane < 16 ? bg[lane] : 1.0f;
  while (s_active) {
    if (lane < 16) {
      const float pr = (1.0f * s_pwm[lane & 3]) * s_pwm[4 + (lane >> 2)];
      float w;
      if (mode == 1) {
        w = cs * pr * __builtin_amdgcn_rcpf(saturation * b + pr);
      } else {
        const float odds = pr / b;
        w = cs / (1 + saturation / odds);
      }
      s_w[lane] = w;
    }
    __syncthreads();
    if (lane == 0) {
      float nw[8];
      for (int c = 0; c < 8; ++c) {
        const int p = c >> 2, a = c & 3;
        if (mode == 2) {
          float acc = 0.0f;
          for (int x = 0; x < 16; ++x)
            if (((x >> (2 * p)) & 3) == a) acc += s_w[x];
          nw[c] = acc;
        } else {
          double acc = 0.0;
          for (int x = 0; x < 16; ++x)
            if (((x >> (2 * p)) & 3) == a) acc += (double)s_w[x];
          nw[c] = (float)acc;
        }
      }
      float change = 0.0f;
      for (int p = 0; p < 2; ++p) {
        float sum = 0.0f;
        for (int a = 0; a < 4; ++a) sum += nw[p * 4 + a];
        for (int a = 0; a < 4; ++a) nw[p * 4 + a] /= sum;
      }
      for (int c = 0; c < 8; ++c) {
        change += fabsf(nw[c] - s_pwm[c]);
        s_pwm[c] = nw[c];
      }
      const int it = state[2 * pw] + 1;
      state[2 * pw] = it;
      s_active = !(change <= threshold || it >= max_it);
      state[2 * pw + 1] = s_active;
      change_out[pw] = change;
    }
    __syncthreads();
  }
  if (lane < 8) old[lane] = s_pwm[lane];
}

template <int W>
int launch_w(pengk_ctx* ctx, int64_t n_pwm, float* d_pwms, float saturation, float threshold, int max_it,
             const uint32_t* d_counts, const float* d_bg, int32_t* d_state, float* d_change) {
#define PENGK_EM_GEO(H, F, PP) launch_geo<W, H, F, PP>(ctx, n_pwm, d_pwms, saturation, threshold, max_it, d_counts, d_bg, d_state, d_change)
  if (ctx->em_fast == 2) return launch_serial<W>(ctx, n_pwm, d_pwms, saturation, threshold, max_it, d_counts, d_bg, d_state, d_change);
  const bool fast = ctx->em_fast != 0;
  // few PWMs: more, smaller workgroups so that every CU gets several waves
  if constexpr (W >= 8) {
    const int64_t wg4 = n_pwm * EmGeo<W, 4>::NB;
    if (wg4 * 4 < (int64_t)ctx->num_cu * 4) return fast ? PENGK_EM_GEO(2, true, 1) : PENGK_EM_GEO(2, false, 1);
    if (wg4 < (int64_t)ctx->num_cu * 8) return fast ? PENGK_EM_GEO(3, true, 1) : PENGK_EM_GEO(3, false, 1);
    // (Sharing table reads between PWMs does not pay: two PWMs per thread need 198 registers -- two waves per SIMD,
    // 4.3 ms against 3.9 ms; two / four PWMs per workgroup as thread groups walking the same slice, so that L1 serves
    // the second read, 4.2 / 4.9 ms; forcing five or more waves per SIMD spills (4.0 / 7.0 / 13 ms).)
  }
  return fast ? PENGK_EM_GEO(4, true, 1) : PENGK_EM_GEO(4, false, 1);
#undef PENGK_EM_GEO
}

}  // namespace

int launch_em(pengk_ctx* ctx, int W, int64_t n_pwm, float* d_pwms, float saturation, float threshold, int max_it,
              const uint32_t* d_counts, const float* d_bg, int32_t* d_state, float* d_change) {
  // (pengk_get_info "em_*": what the chains of THIS call met -- zero for the modes that have no chains)
  if (ctx->d_em_counters) PENGK_HIP(hipMemsetAsync(ctx->d_em_counters, 0, EM_COUNTERS * sizeof(unsigned long long), ctx->stream));
  switch (W) {
    case 2:
      hipLaunchKernelGGL(em_init_kernel, dim3((unsigned)((n_pwm + 255) / 256)), dim3(256), 0, ctx->stream, (int)n_pwm, W, threshold, max_it,
                         d_state, d_change);
      hipLaunchKernelGGL(em_w2_kernel, dim3((unsigned)n_pwm), dim3(64), 0, ctx->stream, d_pwms, d_state, d_change, d_counts, d_bg,
                         saturation, threshold, max_it, ctx->em_fast);
      PENGK_HIP(hipGetLastError());
      return PENGK_OK;
    case 4: return launch_w<4>(ctx, n_pwm, d_pwms, saturation, threshold, max_it, d_counts, d_bg, d_state, d_change);
    case 6: return launch_w<6>(ctx, n_pwm, d_pwms, saturation, threshold, max_it, d_counts, d_bg, d_state, d_change);
    case 8: return launch_w<8>(ctx, n_pwm, d_pwms, saturation, threshold, max_it, d_counts, d_bg, d_state, d_change);
    case 10: return launch_w<10>(ctx, n_pwm, d_pwms, saturation, threshold, max_it, d_counts, d_bg, d_state, d_change);
    case 12: return launch_w<12>(ctx, n_pwm, d_pwms, saturation, threshold, max_it, d_counts, d_bg, d_state, d_change);
    case 14: return launch_w<14>(ctx, n_pwm, d_pwms, saturation, threshold, max_it, d_counts, d_bg, d_state, d_change);
    default: return fail(PENGK_ERR_ARG, "pattern length %d unsupported", W);
  }
}

// (pengk_warmup: loads this translation unit's code object ahead of its first launch)
int warm_em() {
  hipFuncAttributes a;
  PENGK_HIP(hipFuncGetAttributes(&a, (const void*)em_init_kernel));
  return PENGK_OK;
}

}  // namespace pengk

extern "C" int pengk_selftest_division(pengk_ctx* ctx, uint64_t seed, uint32_t pairs_per_thread, uint64_t* h_out) {
  using namespace pengk;
  if (!ctx || !h_out) return fail(PENGK_ERR_ARG, "pengk_selftest_division: NULL argument");
  int rc = enter(ctx);
  if (rc) return rc;
  rc = ensure_scratch(ctx, &ctx->d_misc, &ctx->misc_bytes, 3 * sizeof(unsigned long long));
  if (rc) return rc;
  PENGK_HIP(hipMemsetAsync(ctx->d_misc, 0, 3 * sizeof(unsigned long long), ctx->stream));
  hipLaunchKernelGGL(em_div_check_kernel, dim3(4096), dim3(256), 0, ctx->stream, (unsigned long long)seed, pairs_per_thread,
                     (unsigned long long*)ctx->d_misc);
  PENGK_HIP(hipGetLastError());
  PENGK_HIP(hipMemcpyAsync(h_out, ctx->d_misc, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  PENGK_HIP(hipStreamSynchronize(ctx->stream));
  return PENGK_OK;
}
