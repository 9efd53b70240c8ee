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
#include <algorithm>
#include <type_traits>

#include "pengk_internal.h"
#include "seqsum.h"

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

// HIMAX = 4: 256 leaves per thread (fewest partial products; best when the grid is full anyway).
// HIMAX = 3 / 2: 64 / 16 leaves per thread, 4x / 16x more workgroups -- for small PWM batches that would
// otherwise leave most CUs with a single wave (the 16-PWM batch of a typical run takes HIMAX = 3 at W = 10:
// 1024 workgroups, four per CU, and a fourth of the per-workgroup reductions of HIMAX = 2).
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
// The reference adds the 4^W weights into each PWM cell one after the other in float32 (src/peng.cpp:121-127);
// the motifs' merge and redundancy decisions downstream compare similarity scores that are exactly tied in real
// arithmetic for reverse-complement twins, so the last bits of those sums decide what the program prints.
// A cell's sum is inherently sequential, but cells and PWMs are independent and the weights are not:
//   em_weights_kernel    all weights w(x) of a PWM in parallel (reference float operations), to a scratch table;
//   em_fold_scan_kernel  (W >= 8) one wave per cell evaluates the cell's chain of roundings as a scan (seqsum.h);
//   em_fold_kernel       (W <= 6, or option em_serial_scan = 0) one workgroup per position and PWM:
//                        the four cells (p, a) walk THEIR terms -- the x whose digit p is a, ascending -- from LDS,
//                        where loader waves stage them with coalesced loads, one dependent addition after the other.
// ---------------------------------------------------------------------------------------------
template <int W, bool T0>
__global__ __launch_bounds__(256) void em_weights_kernel(const float* __restrict__ pwms, const int32_t* __restrict__ state,
                                                         const uint32_t* __restrict__ counts, const float* __restrict__ bg,
                                                         float saturation, float* __restrict__ wbuf,
                                                         uint32_t* __restrict__ bad) {
  const int pw = blockIdx.y;
  if (state[2 * pw + 1] == 0) return;
  __shared__ float s_pwm[W * 4];
  if (threadIdx.x < W * 4) s_pwm[threadIdx.x] = pwms[(size_t)pw * W * 4 + threadIdx.x];
  __syncthreads();
  const uint32_t np = 1u << (2 * W);
  float* out = wbuf + (size_t)pw * ((size_t)np << (T0 ? 1 : 0));
  // A thread takes the 16 x that share their low W-2 digits: the product over those positions is built once, in the
  // reference's order ((1*pwm[0][x0])*pwm[1][x1])... (src/peng.cpp:180-197), and the last two factors are applied per x
  // -- the same multiplications in the same order for every x, 1.25 per x instead of W (the ten LDS look-ups and
  // multiplications were 40 % of this kernel; the rest is its three IEEE divisions).
  constexpr uint32_t NLOW = 1u << (2 * W - 4);
  float hi0[4], hi1[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    hi0[a] = s_pwm[(W - 2) * 4 + a];
    hi1[a] = s_pwm[(W - 1) * 4 + a];
  }
  for (uint32_t low = blockIdx.x * blockDim.x + threadIdx.x; low < NLOW; low += gridDim.x * blockDim.x) {
    float pl = 1.0f;
#pragma unroll
    for (int p = 0; p < W - 2; ++p) pl = pl * s_pwm[p * 4 + ((low >> (2 * p)) & 3u)];
#pragma unroll
    for (uint32_t a8 = 0; a8 < 4u; ++a8) {
      const float p8 = pl * hi0[a8];
#pragma unroll
      for (uint32_t a9 = 0; a9 < 4u; ++a9) {
        const uint32_t x = low | (a8 << (2 * W - 4)) | (a9 << (2 * W - 2));
        const float pr = p8 * hi1[a9];
        const float odds = pr / bg[x];
        const float v = ((float)counts[x] * saturation) / (1 + saturation / odds);  // :124-125
        out[x] = v;
        // position 0's cells take every fourth x: a second, permuted copy of the table with the four cells' terms
        // contiguous (term c of cell a at np + a 4^(W-1) + c) lets the scan fetch them like the cells of position W-1
        // (otherwise each of the four cells moves every line and issues four times the loads -- they were the last to
        // finish)
        if (T0) out[np + (x & 3u) * (np / 4u) + (x >> 2)] = v;
        // a negative or non-finite weight (degenerate PWM / background entries): this PWM's cells are summed by the
        // plain loop of the finalize kernel instead of the scan (seqsum.h)
        if (__float_as_uint(v) > 0x7F7FFFFFu) bad[pw] = 1u;
      }
    }
  }
}

// One workgroup per (position p, PWM): the four cells (p, a) of a position partition the table -- every x has exactly
// one digit at position p -- so the workgroup streams each cell's terms, in the cell's order, through LDS:
//   waves 1, 2  (loaders) fetch chunk s + 2 of the four term streams with coalesced 16-byte loads (a cell's terms are
//               runs of 4^p consecutive x: whole cache lines per request instead of one line per lane and load, which
//               held the first versions of this kernel at 3.8 and 1.9 ms per iteration), and put chunk s + 1, fetched
//               during the previous stage, into the other LDS buffer;
//   wave 0      (lanes 0..3 = a) adds chunk s from LDS, strictly in order: this IS the reference's rounding sequence.
// A cell's chain is 4^(W-1) dependent float32 additions, about one per issue turn of its wave: what is left is that
// chain (0.26 M additions at W = 10) plus one LDS read per four terms.  Cells and PWMs are independent: 10 x n_pwm
// workgroups fill the chip from 26 PWMs on.
template <int W>
struct FoldGeo {
  static constexpr uint32_t TERMS = 1u << (2 * W - 2);             // per cell
  static constexpr uint32_t C = TERMS < 1024u ? TERMS : 1024u;     // terms per cell and stage
  static constexpr uint32_t STAGES = TERMS / C;
  static constexpr uint32_t QUADS = C;                             // 16-byte pieces per stage: 4 cells x C / 4
  static constexpr uint32_t LOADERS = 128;
  static constexpr uint32_t QPT = (QUADS + LOADERS - 1) / LOADERS;  // quads per loader thread and stage
  static constexpr uint32_t ROW = C + 68;                          // floats per cell row in LDS: + 16 quads the adder may read past the
                                                                   // end (never added), + 16 B (rows on different banks)
};

template <int W>
__global__ __launch_bounds__(192) void em_fold_kernel(const int32_t* __restrict__ state, const float* __restrict__ wbuf,
                                                      double* __restrict__ partials, uint32_t pwm_stride) {
  using F = FoldGeo<W>;
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int pw = blockIdx.y;
  if (state[2 * pw + 1] == 0) return;
  const uint32_t p = blockIdx.x;  // position
  const uint32_t np = 1u << (2 * W);
  const float* w = wbuf + (size_t)pw * pwm_stride;
  __shared__ __attribute__((aligned(16))) float buf[2][4][F::ROW];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const uint32_t run = 1u << (2 * p);  // a cell's terms come in runs of 4^p consecutive x, one run per 4^(p+1)

  // loader thread: quad q of a stage = terms 4 (q % (C/4)) .. + 3 of cell a = q / (C/4)
  f4 pend[F::QPT];
  const uint32_t lt = threadIdx.x - 64u;  // loader index (waves 1, 2)
  auto fetch = [&](uint32_t stage) {
#pragma unroll
    for (uint32_t i = 0; i < F::QPT; ++i) {
      const uint32_t q = lt + i * F::LOADERS;
      if (q < F::QUADS) {
        if (p == 0) {  // position 0: term t of cell a is x = 4 t + a -- quad q holds term q of all four cells
          pend[i] = *reinterpret_cast<const f4*>(w + 4u * (stage * F::C + q));
        } else {
          const uint32_t a = q / (F::C / 4u), t = stage * F::C + 4u * (q % (F::C / 4u));
          const uint32_t x = ((t >> (2 * p)) << (2 * p + 2)) | (a << (2 * p)) | (t & (run - 1u));
          pend[i] = *reinterpret_cast<const f4*>(w + x);
        }
      }
    }
  };
  auto deposit = [&](uint32_t b) {
#pragma unroll
    for (uint32_t i = 0; i < F::QPT; ++i) {
      const uint32_t q = lt + i * F::LOADERS;
      if (q < F::QUADS) {
        if (p == 0) {
          buf[b][0][q] = pend[i].x;
          buf[b][1][q] = pend[i].y;
          buf[b][2][q] = pend[i].z;
          buf[b][3][q] = pend[i].w;
        } else {
          const uint32_t a = q / (F::C / 4u), j = 4u * (q % (F::C / 4u));
          *reinterpret_cast<f4*>(&buf[b][a][j]) = pend[i];
        }
      }
    }
  };

  if (wave != 0) {
    fetch(0);
    deposit(0);
    if (F::STAGES > 1) fetch(1);
  }
  __syncthreads();
  float acc = 0.0f;
#pragma unroll 1
  for (uint32_t s = 0; s < F::STAGES; ++s) {
    if (wave != 0) {
      if (s + 1 < F::STAGES) deposit((s + 1) & 1u);  // fetched during the previous stage
      if (s + 2 < F::STAGES) fetch(s + 2);
    } else if (lane < 4u) {
      // two register sets of 16 quads: the LDS reads of the next 64 terms are in flight while these 64 are added
      const f4* src = reinterpret_cast<const f4*>(&buf[s & 1u][lane][0]);
      constexpr uint32_t NQ = F::C / 4u, G = NQ < 16u ? NQ : 16u;
      static_assert(NQ % (2u * G) == 0u || NQ == G, "quads per stage");
      f4 va[G], vb[G];
      auto rd = [&](f4 (&v)[G], uint32_t i0) {
#pragma unroll
        for (uint32_t k = 0; k < G; ++k) v[k] = src[i0 + k];
      };
      auto add = [&](f4 (&v)[G]) {
        // all quads of the set are "used" here at once: ONE s_waitcnt in front of the 4 G additions instead of one per
        // quad (every instruction of the adding wave, waits included, costs the chain an issue turn)
        if constexpr (G == 16)
          asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                       "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
#pragma unroll
        for (uint32_t k = 0; k < G; ++k) {
          acc += v[k].x;
          acc += v[k].y;
          acc += v[k].z;
          acc += v[k].w;
        }
      };
      rd(va, 0);
      if constexpr (NQ == G) {
        add(va);
      } else {
#pragma unroll 1
        for (uint32_t i = 0; i < NQ; i += 2u * G) {
          // (scheduling barriers: left alone, the compiler moves each group of reads behind the additions in front
          // of it and waits for every quad right after asking for it)
          rd(vb, i + G);
          __builtin_amdgcn_sched_barrier(0);
          add(va);
          __builtin_amdgcn_sched_barrier(0);
          rd(va, i + 2u * G);  // unconditional (a branch here costs 32 register moves per turn): the last turn reads
          __builtin_amdgcn_sched_barrier(0);
          add(vb);             // the row's padding and never adds it
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    __syncthreads();
  }
  if (wave == 0 && lane < 4u) partials[(size_t)pw * (W * 4) + p * 4u + lane] = (double)acc;  // layout of EmGeo<W, 16>
}

// The same sums -- the same roundings, seqsum.h -- by one wave per cell: a cell's 4^(W-1) terms in blocks of 4096, each
// block fetched with coalesced loads, spread over 64 LDS rows of 64 consecutive terms, and evaluated as 64 stretches
// that the wave composes.  The chain is walked in 4^(W-1) / 4096 steps of ~1.4 us instead of 4^(W-1) dependent
// additions (W = 10, 16 PWMs: 0.71 -> 0.09 ms per iteration), and a batch of PWMs fills the chip with 4 W waves per PWM.
// What is left of a step is one wave's own dependent work: 128 additions, the prefix composition, the wait for LDS.
template <int W>
struct EmTerms {
  typedef seqsum::f4 f4;
  const float* __restrict__ w;  // the PWM's weight table (x order)
  uint32_t p, a;                // the cell: terms are the x whose digit p >= 1 is a, ascending
  // term c of the cell is x = [c's digits p.. | a | c's digits 0..p-1]
  __device__ __forceinline__ uint32_t x_of(uint32_t c) const {
    return ((c >> (2u * p)) << (2u * p + 2u)) | (a << (2u * p)) | (c & ((1u << (2u * p)) - 1u));
  }
  // runs of 4^p >= 4 consecutive x: 16-byte loads, terms 256 k + 4 lane .. + 3 of the block in R[4 k ..].
  // x_of(4096 b + r) = F(b) + x_of(r) for r < 4096 (4096 b is a multiple of 4^p, or 4^p a multiple of 4096: no carry
  // between the two parts of c): the per-lane part, sixteen byte offsets, is computed once (bind), the per-block part
  // is a scalar -- a load costs no vector instruction (computing x_of per load was a quarter of a step).
  uint32_t g[16];
  __device__ __forceinline__ void bind(uint32_t lane) {
#pragma unroll
    for (uint32_t k = 0; k < 16u; ++k) g[k] = 4u * x_of(256u * k + 4u * lane);
  }
  // share `part` of NF: the loads k = part * 16 / NF .. of the block (part is wave-uniform; the selects below pick the
  // share's offsets once per call)
  template <uint32_t NF>
  __device__ __forceinline__ void load(uint32_t b, uint32_t part, uint32_t lane, float (&R)[64 / NF]) const {
    // wave-uniform (readfirstlane: the base stays in scalar registers, the load takes it plus a 32-bit lane offset)
    const uint32_t F = (uint32_t)__builtin_amdgcn_readfirstlane((int)(x_of(b * seqsum::BLOCK) - (a << (2u * p))));
    const char* base = reinterpret_cast<const char*>(w + F);
#pragma unroll
    for (uint32_t k = 0; k < 16u / NF; ++k) {
      uint32_t off = g[k];
#pragma unroll
      for (uint32_t q = 1; q < NF; ++q) off = part == q ? g[q * (16u / NF) + k] : off;
      const f4 v = *reinterpret_cast<const f4*>(base + off);
      R[4u * k] = v.x;
      R[4u * k + 1u] = v.y;
      R[4u * k + 2u] = v.z;
      R[4u * k + 3u] = v.w;
    }
  }
  template <uint32_t NF>
  __device__ __forceinline__ void deposit(uint32_t part, uint32_t lane, const float (&R)[64 / NF], float* lds) const {
    float* dst = lds + (4u * (16u / NF) * part + (lane >> 4)) * seqsum::SEG_STRIDE + 4u * (lane & 15u);
#pragma unroll
    for (uint32_t k = 0; k < 16u / NF; ++k) {
      f4 v;
      v.x = R[4u * k];
      v.y = R[4u * k + 1u];
      v.z = R[4u * k + 2u];
      v.w = R[4u * k + 3u];
      *reinterpret_cast<f4*>(dst + 4u * k * seqsum::SEG_STRIDE) = v;
    }
  }
  __device__ __forceinline__ float serial() const { return 0.0f; }  // (unused: flagged PWMs never reach the scan)

  // Block b straight into LDS (global_load_lds: no registers, so several blocks can be on their way and the wait for one
  // of them is a counted s_waitcnt; seqsum.h, walk_chain).  The k-th of the sixteen loads writes 1 KiB of LDS in lane
  // order: four rows of 64 terms, lane l the 16-byte slot l & 15 of row 4 k + (l >> 4) -- and which four terms lie there
  // is the reader's choice: slot c of row r holds the terms 4 (c ^ (r & 15)) .. + 3 of the row, so that the lanes that
  // read one slot number of their own rows together hit sixteen different slots (seqsum::Row::read_staged).
  static constexpr uint32_t STAGE_LOADS = 16;
  uint32_t gs[16];
  __device__ __forceinline__ void bind_stage(uint32_t lane) {
#pragma unroll
    for (uint32_t k = 0; k < 16u; ++k) {
      const uint32_t r = 4u * k + (lane >> 4);
      gs[k] = 4u * x_of(64u * r + 4u * ((lane & 15u) ^ (r & 15u)));
    }
  }
  __device__ __forceinline__ void stage(uint32_t b, uint32_t /*lane*/, seqsum::lds_float* buf) const {
    const uint32_t F = (uint32_t)__builtin_amdgcn_readfirstlane((int)(x_of(b * seqsum::BLOCK) - (a << (2u * p))));
    const char* base = reinterpret_cast<const char*>(w + F);
#pragma unroll
    for (uint32_t k = 0; k < 16u; ++k)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + gs[k]),
                                       (__attribute__((address_space(3))) void*)(buf + 256u * k), 16, 0, 0);
  }
};

// The four cells of position 0 take every fourth float: term c of cell (0, a) is x = 4 c + a.  Two ways to feed them:
//  * a second, permuted copy of the table that the weights kernel writes beside it (term c of cell a at
//    np + a 4^(W-1) + c): the cells then fetch like those of position W-1.  Twice the table bytes per PWM.  W <= 10,
//    where the tables of a batch stay in the Infinity Cache and a step costs what the evaluating wave costs;
//  * straight from the table (EmTerms0): a block is the 16384 x from 16384 b, sixty-four dword loads per lane (lane l
//    takes x = 4 (64 k + l) + a: each wave-load walks 1 KiB of consecutive lines and keeps a quarter of it; the four
//    cells run side by side on one XCD and share the lines in its L2).  Four times the load instructions, and at
//    W = 10 these four cells then finish last (0.95 -> 1.17 ms for 16 PWMs, 49 -> 56 ms for 1000); but at W = 12, where
//    every table byte comes from HBM (128 MiB per PWM with the copy), half the bytes win: 25.1 -> 21.8 ms for 16 PWMs
//    x 10 iterations (profiles/r03_em_experiments.log).  W >= 12.
template <int W>
struct ScanCopy0 {
  static constexpr bool value = W <= 10;
};
template <int W>
struct EmTerms0 {
  const float* __restrict__ w;  // the PWM's weight table (x order)
  uint32_t a;
  template <uint32_t NF>
  __device__ __forceinline__ void load(uint32_t b, uint32_t /*part*/, uint32_t lane, float (&R)[64 / NF]) const {
    static_assert(NF == 1u, "whole blocks");
    const float* base = w + (size_t)b * (4u * seqsum::BLOCK) + 4u * lane + a;
#pragma unroll
    for (uint32_t k = 0; k < 64u; ++k) R[k] = base[256u * k];
  }
  template <uint32_t NF>
  __device__ __forceinline__ void deposit(uint32_t /*part*/, uint32_t lane, const float (&R)[64 / NF], float* lds) const {
#pragma unroll
    for (uint32_t k = 0; k < 64u; ++k) lds[k * seqsum::SEG_STRIDE + lane] = R[k];
  }
  __device__ __forceinline__ float serial() const { return 0.0f; }  // (unused: flagged PWMs never reach the scan)

  // (as EmTerms::stage, with dword loads: the k-th of 64 writes row k, lane l the term whose place is l)
  static constexpr uint32_t STAGE_LOADS = 64;
  __device__ __forceinline__ void bind_stage(uint32_t) {}
  __device__ __forceinline__ void stage(uint32_t b, uint32_t lane, seqsum::lds_float* buf) const {
    const float* base = w + (size_t)b * (4u * seqsum::BLOCK) + a;
#pragma unroll
    for (uint32_t k = 0; k < 64u; ++k) {
      const uint32_t term = ((((lane >> 2) ^ (k & 15u)) << 2) | (lane & 3u));  // slot (l >> 2) of row k holds slot ^ (k & 15)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + 4u * (64u * k + term)),
                                       (__attribute__((address_space(3))) void*)(buf + 64u * k), 4, 0, 0);
    }
  }
};

// Workgroup -> (PWM, cell): consecutive workgroups go to consecutive XCDs (8 on gfx950, each with its own 4 MiB L2), so
// the 4 W cells of a PWM are given to ONE XCD: a PWM's weight table (4^W floats, 4 MiB at W = 10) is read once per
// position, and the cells of positions 0 .. W-3 walk it side by side within a 256 KiB window -- from that XCD's L2
// instead of W times across the fabric (the scan is bound by those reads, not by its arithmetic).  The grid is padded to
// whole groups of 8 PWMs; workgroups of the padding leave at once.
template <int W>
__global__ __launch_bounds__(seqsum::CHAIN_THREADS) void em_fold_scan_kernel(const int32_t* __restrict__ state, const float* __restrict__ wbuf,
                                                          double* __restrict__ partials, const uint32_t* __restrict__ bad,
                                                          uint32_t n_pwm) {
  static_assert((1u << (2 * W - 2)) % seqsum::BLOCK == 0u, "whole blocks per cell");
  const uint32_t lin = blockIdx.x + gridDim.x * blockIdx.y, slot = lin >> 3;
  const uint32_t cell = slot % (4u * W), pw = (lin & 7u) + 8u * (slot / (4u * W));
  if (pw >= n_pwm || state[2 * pw + 1] == 0 || bad[pw]) return;
#ifndef PENGK_SCAN_LDS_PAD
#define PENGK_SCAN_LDS_PAD 0
#endif
  __shared__ __attribute__((aligned(16))) float lds[seqsum::CHAIN_LDS_FLOATS + PENGK_SCAN_LDS_PAD];
  constexpr uint32_t NP = 1u << (2 * W);
  constexpr uint32_t NBLK = (1u << (2 * W - 2)) / seqsum::BLOCK;
  float s;
  if constexpr (ScanCopy0<W>::value) {
    // position 0 reads the weights kernel's second copy, where its four cells lie like those of position W-1
    const float* w = wbuf + (size_t)pw * 2u * NP;
    EmTerms<W> src = (cell >> 2) == 0u ? EmTerms<W>{w + NP, (uint32_t)(W - 1), cell & 3u} : EmTerms<W>{w, cell >> 2, cell & 3u};
    src.bind(threadIdx.x & 63u);
    s = seqsum::fold_chain<EmTerms<W>, false>(src, NBLK, lds, threadIdx.x);
  } else {
    const float* w = wbuf + (size_t)pw * NP;
    if ((cell >> 2) == 0u) {  // (block-uniform)
      const EmTerms0<W> src0{w, cell & 3u};
      s = seqsum::fold_chain<EmTerms0<W>, false>(src0, NBLK, lds, threadIdx.x);
    } else {
      EmTerms<W> src{w, cell >> 2, cell & 3u};
      src.bind(threadIdx.x & 63u);
      s = seqsum::fold_chain<EmTerms<W>, false>(src, NBLK, lds, threadIdx.x);
    }
  }
  if (threadIdx.x == 0) partials[(size_t)pw * (W * 4) + cell] = (double)s;  // cell (p, a) = 4 p + a: layout of EmGeo<W, 16>
}

// One block per PWM: sum the per-block partials in block order, then the reference's float32
// epilogue: normalise rows (:129), change = sum |new - old| (:132-137), swap (:140-143).
// (The work of one PWM, by a workgroup of at least 4 W threads: em_finalize_kernel, or the last of a PWM's cells in
// em_chain_kernel.)
template <int W, int HIMAX>
__device__ __forceinline__ void finalize_pwm(int pw, float* __restrict__ pwms, int32_t* __restrict__ state,
                                             float* __restrict__ change_out, const double* __restrict__ partials, float threshold,
                                             int max_it, uint32_t* __restrict__ bad, const float* __restrict__ wbuf, uint32_t pwm_stride,
                                             float* s_new) {
  using G = EmGeo<W, HIMAX>;
  const int e = threadIdx.x;
  // serial mode with the scan: a PWM the weights kernel flagged (a negative or non-finite weight -- degenerate inputs
  // only) was left out by em_fold_scan_kernel; its cells are summed here, one thread per cell, by the plain loop
  const bool flagged = bad && bad[pw];
  if (e < G::CELLS) {
    if (flagged) {
      const float* w = wbuf + (size_t)pw * pwm_stride;
      const uint32_t p = (uint32_t)e >> 2, a = (uint32_t)e & 3u;
      float acc = 0.0f;
      for (uint32_t c = 0; c < (1u << (2 * W - 2)); ++c)
        acc += w[((c >> (2u * p)) << (2u * p + 2u)) | (a << (2u * p)) | (c & ((1u << (2u * p)) - 1u))];
      s_new[e] = acc;
    } else {
      const double* src = partials + (size_t)pw * G::NB * G::CELLS + e;
      double v = 0.0;
      // (device-scope loads: in em_chain_kernel the values were written by other workgroups of the same launch)
      for (int b = 0; b < G::NB; ++b) v += __hip_atomic_load(src + (size_t)b * G::CELLS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_new[e] = (float)v;
    }
  }
  __syncthreads();
  if (bad && e == 0) bad[pw] = 0u;  // the next iteration's weights set it again if need be
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

template <int W, int HIMAX>
__global__ __launch_bounds__(64) void em_finalize_kernel(float* __restrict__ pwms, int32_t* __restrict__ state,
                                                         float* __restrict__ change_out, const double* __restrict__ partials,
                                                         float threshold, int max_it, uint32_t* __restrict__ bad,
                                                         const float* __restrict__ wbuf, uint32_t pwm_stride) {
  const int pw = blockIdx.x;
  if (state[2 * pw + 1] == 0) return;
  __shared__ float s_new[W * 4];
  finalize_pwm<W, HIMAX>(pw, pwms, state, change_out, partials, threshold, max_it, bad, wbuf, pwm_stride, s_new);
}

// ---- the scan with its blocks evaluated ahead of the chain (seqsum.h, "blocks ahead of their chain"; W >= 10) ----------
// Per iteration and PWM:
//   em_weights_span_kernel   the weights, span by span, and with them the plain sum of every block of every cell;
//   (em_block_predict_kernel their prefix per cell = an estimate of the sum in front of each block; a block whose estimate
//                            stays clear of a power of two from its first to its last term gets that binade -- for cells of
//                            up to 1024 blocks the next kernel does that itself)
//   em_span_eval_kernel      every block with a binade gets its two increments (block_increments) -- all blocks of all
//                            cells at once, a workgroup per span of the table, instead of one after the other per cell;
//   em_chain_kernel          one wave per cell walks the blocks: one addition per evaluated block, fold_block for the
//                            others (the first block, where the sum climbs from zero, and the few where it crosses).
// A span = 16384 consecutive x = 4^7: for a position p <= 6 a span holds block `span` of each of the four cells (p, a);
// for p >= 7 it holds four consecutive blocks of the one cell (p, digit p of the span).
template <int W>
struct BlockGeo {
  static_assert(W >= 8, "a span is 4^7 x");
  static constexpr uint32_t NP = 1u << (2 * W);
  static constexpr uint32_t SPANS = NP >> 14;
  static constexpr uint32_t NBLK = (1u << (2 * W - 2)) / seqsum::BLOCK;  // per cell (= SPANS)
  static constexpr uint32_t CELLS = 4u * W;
  // cells of up to 1024 blocks: the evaluating wave adds up the cell's block sums itself; longer ones (W = 14) get their
  // prefix from em_block_predict_kernel
  static constexpr bool PREDICT_IN_EVAL = NBLK <= 1024u;
  // the block of cell (p, a) that quarter q of span sp belongs to (p >= 7), and the cell's a
  static __device__ __forceinline__ uint32_t high_block(uint32_t p, uint32_t sp, uint32_t q) {
    const uint32_t sh = 2u * (p - 7u);
    return 4u * (((sp >> (sh + 2u)) << sh) | (sp & ((1u << sh) - 1u))) + q;
  }
  static __device__ __forceinline__ uint32_t high_digit(uint32_t p, uint32_t sp) { return (sp >> (2u * (p - 7u))) & 3u; }
};

// The binade of a block from the estimates of the sum in front of it and behind it, or NO_BINADE when the two -- widened
// by 2^-9, far more than a float32 chain of 4^13 terms drifts from the exact sum in practice -- do not share one.  A wrong
// guess costs time, never the result (seqsum.h).
__device__ __forceinline__ uint32_t block_binade(double before, double after, uint32_t skew = 0u, uint32_t key = 0u) {
  const float lo = (float)(before * (1.0 - 1.0 / 512.0)), hi = (float)(after * (1.0 + 1.0 / 512.0));
  const bool sane = lo >= 0.0f && hi < __uint_as_float(0x7F000000u);
  uint32_t e = sane && seqsum::binade_of(lo) == seqsum::binade_of(hi) ? seqsum::binade_of(lo) : seqsum::NO_BINADE;
  // Test hook (option "em_test_skew" = n > 0): about every n-th block gets a WRONG answer -- the binade above the
  // right one, or a binade where there is none to be had -- so that the suite exercises what a bad estimate costs
  // (the chain's checks, its fetches on demand) far more often than real inputs do.  Results must not change.
  if (skew != 0u && sane && ((key * 2654435761u) >> 16) % skew == 0u) {
    if (e == seqsum::NO_BINADE) e = seqsum::binade_of(lo);
    else if (e < 200u) e += 1u;
  }
  return e;
}

// ---- the three IEEE divisions of a weight, without the range scaling when it cannot matter ---------------------------
// `a / b` in float compiles to v_div_scale x 2, v_rcp, five fma / mul, v_div_fmas, v_div_fixup (11 instructions, the
// reciprocal at quarter rate): 33 of the ~47 vector instructions of a weight.  v_div_scale returns its operand unchanged
// and VCC = 0 -- so that v_div_fmas is a plain fma -- and v_div_fixup passes the quotient through, when (gfx9 ISA,
// V_DIV_SCALE_F32 / V_DIV_FIXUP_F32): numerator and denominator are finite, the denominator is normal and below 2^126,
// the numerator's biased exponent is above 23 (or the numerator is zero: every product below is then zero, and so is the
// fixup's answer), the exponents differ by less than 96 and the quotient is normal.  Then the eight instructions in
// between ARE the division, bit for bit: the same v_rcp_f32, the same fmas in the same order.  lean_div issues exactly
// those.  Whether a workgroup may use it is decided once per workgroup from the RANGES its operands can take --
// the PWM's columns give the range of the product, em_bg_range_kernel the range of the background table, the count
// table's 32 bits the range of c * s -- with a factor of two of slack on every derived bound (lean_ranges_ok);
// a workgroup whose ranges do not qualify (tiny PWM entries, a degenerate background) runs the plain divisions.
__device__ __forceinline__ float lean_div(float a, float b) {
  const float y0 = __builtin_amdgcn_rcpf(b);
  const float e0 = __builtin_fmaf(-b, y0, 1.0f);
  const float y1 = __builtin_fmaf(e0, y0, y0);
  const float q0 = a * y1;
  const float r0 = __builtin_fmaf(-b, q0, a);
  const float q1 = __builtin_fmaf(r0, y1, q0);
  const float r1 = __builtin_fmaf(-b, q1, a);
  return __builtin_fmaf(r1, y1, q1);
}
template <bool LEAN>
__device__ __forceinline__ float em_div(float a, float b) {
  if constexpr (LEAN) return lean_div(a, b);
  else return a / b;
}
// a / b for every a in [a_lo, a_hi] (or a == 0) and b in [b_lo, b_hi], all bounds positive: is the unscaled sequence the
// division?  (biased exponents; the quotient of a and b lies in [2^(Ea - Eb - 1), 2^(Ea - Eb + 1)))
__device__ __forceinline__ bool lean_div_ok(float a_lo, float a_hi, float b_lo, float b_hi) {
  auto fin = [](float x) { return __float_as_uint(x) - 0x00800000u < 0x7F000000u; };  // normal, finite, positive
  if (!(fin(a_lo) && fin(a_hi) && fin(b_lo) && fin(b_hi)) || a_lo > a_hi || b_lo > b_hi) return false;
  const int ea_lo = (int)(__float_as_uint(a_lo) >> 23), ea_hi = (int)(__float_as_uint(a_hi) >> 23);
  const int eb_lo = (int)(__float_as_uint(b_lo) >> 23), eb_hi = (int)(__float_as_uint(b_hi) >> 23);
  return eb_hi <= 251 && ea_lo >= 25 && ea_hi - eb_lo <= 94 && ea_lo - eb_hi >= -123;
}
// The ranges of one PWM's three divisions (src/peng.cpp:124-125, 180-197): odds = pr / bg, t = s / odds,
// w = (c s) / (1 + t).  bg_range = {min, max} of the background table as float bits (em_bg_range_kernel).
template <int W>
__device__ __forceinline__ bool lean_ranges_ok(const float* s_pwm, uint32_t bg_lo_bits, uint32_t bg_hi_bits, float saturation) {
  float p_lo = 1.0f, p_hi = 1.0f;
  for (int p = 0; p < W; ++p) {
    const float a = s_pwm[p * 4], b = s_pwm[p * 4 + 1], c = s_pwm[p * 4 + 2], d = s_pwm[p * 4 + 3];
    if (!(a > 0.0f && b > 0.0f && c > 0.0f && d > 0.0f)) return false;
    p_lo *= fminf(fminf(a, b), fminf(c, d));
    p_hi *= fmaxf(fmaxf(a, b), fmaxf(c, d));
  }
  // (products round: half a unit in the last place per factor, far inside the factor of two below)
  p_lo *= 0.5f;
  p_hi *= 2.0f;
  const float b_lo = __uint_as_float(bg_lo_bits), b_hi = __uint_as_float(bg_hi_bits);
  if (bg_hi_bits > 0x7F7FFFFFu || !(saturation > 0.0f)) return false;  // a negative or non-finite background entry
  if (!lean_div_ok(p_lo, p_hi, b_lo, b_hi)) return false;
  const float o_lo = p_lo / b_hi * 0.5f, o_hi = p_hi / b_lo * 2.0f;  // odds
  if (!lean_div_ok(saturation, saturation, o_lo, o_hi)) return false;
  const float t_hi = saturation / o_lo * 2.0f;  // s / odds <= t_hi; 1 + t in [1, 2 (1 + t_hi)]
  const float n_lo = saturation * 0.5f, n_hi = saturation * 8589934592.0f;  // c s, c in [1, 2^32): [s / 2, 2^33 s]
  return lean_div_ok(n_lo, n_hi, 1.0f, (1.0f + t_hi) * 2.0f);
}

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

// What F_j, the finalize step of iteration j, makes of a PWM: the reference's float32 epilogue -- normalise rows
// (src/peng.cpp:129, src/iupac_pattern.cpp:291-303), change = sum |new - old| in p-major order (:132-137) -- from the cell
// sums in s_new[0 .. 4 W) and the previous PWM in s_old[0 .. 4 W) (LDS; every thread of the workgroup calls it between
// two barriers of its own).  The new PWM is left in s_new, the cells' |new - old| in s_old; returns `change` (the same
// value in every thread: each adds up the 4 W differences itself, in p-major order, from 16-byte LDS reads).
template <int W>
__device__ __forceinline__ float finalize_rows(float* s_new, float* s_old, uint32_t t) {
  typedef seqsum::f4 f4;
  float mine = 0.0f, diff = 0.0f;
  if (t < 4u * W) {
    const f4 row = reinterpret_cast<const f4*>(s_new)[t >> 2];
    float sum = 0.0f;
    sum += row.x;
    sum += row.y;
    sum += row.z;
    sum += row.w;
    mine = s_new[t] / sum;
    diff = fabsf(mine - s_old[t]);
  }
  __syncthreads();
  if (t < 4u * W) {
    s_new[t] = mine;
    s_old[t] = diff;
  }
  __syncthreads();
  float change = 0.0f;
  f4 d[W];
#pragma unroll
  for (int p = 0; p < W; ++p) d[p] = reinterpret_cast<const f4*>(s_old)[p];
#pragma unroll
  for (int p = 0; p < W; ++p) {
    change += d[p].x;
    change += d[p].y;
    change += d[p].z;
    change += d[p].w;
  }
  return change;
}

// The pieces of the per-PWM state the two-launch scheme keeps beside the caller's arrays (all indexed by PWM):
//   run[2][n]     run[j & 1] = "still running" behind F_j; launch k reads run[k & 1] (= behind F_(k-2)) and writes
//                 run[(k - 1) & 1]; the chains of launch k read what it wrote.  Never read and written by one launch.
//   pwm1[n][4 W]  PWM_j for odd j (even j: the caller's array): launch k reads PWM_(k-2), writes PWM_(k-1) to the other one.
//   bad[2][n]     bad[k & 1] = "launch k met a weight the scan cannot take" (read by its chains); launch k clears the other.
struct FusedState {
  uint32_t* run;       // [2][n]
  float* pwm0;         // the caller's PWMs (PWM_j, j even)
  float* pwm1;         // scratch (j odd)
  uint32_t* bad;       // [2][n]
  const float* cellsum;  // [n][4 W]: what the chains of the previous launch left
  int32_t* state;      // the caller's {iterations, running} pairs
  float* change;       // the caller's `change`
  uint32_t n;          // PWMs of this batch
  uint32_t run_stride, bad_stride;  // words between the two copies of run[] / bad[]
};

// Common head of every workgroup of em_span_fused_kernel: F_(k-1) for PWM pw, or PWM_0 at k = 1.  Leaves the PWM the
// weights are to be computed from in s_pwm and returns whether the PWM is still running.  `writer`: this workgroup
// records the step (exactly one workgroup per PWM and launch).
template <int W>
__device__ __forceinline__ bool fused_head(const FusedState& fs, uint32_t pw, uint32_t k, float threshold, int max_it, bool writer,
                                           float* s_pwm, float* s_old, uint32_t t) {
  constexpr uint32_t CELLS = 4u * W;
  const uint32_t was_running = fs.run[(size_t)(k & 1u) * fs.run_stride + pw];
  const float* prev = (k >= 2u && (k & 1u)) ? fs.pwm1 : fs.pwm0;  // PWM_(k-2) (k = 1: PWM_0)
  float old = 0.0f, sum = 0.0f;
  if (t < CELLS) {
    old = prev[(size_t)pw * CELLS + t];
    if (k >= 2u) sum = fs.cellsum[(size_t)pw * CELLS + t];
  }
  if (!was_running) {  // (workgroup-uniform)
    if (writer && t == 0) {
      fs.run[(size_t)((k - 1u) & 1u) * fs.run_stride + pw] = 0u;
      fs.bad[(size_t)((k + 1u) & 1u) * fs.bad_stride + pw] = 0u;
    }
    return false;
  }
  if (k < 2u) {
    if (t < CELLS) s_pwm[t] = old;
    if (writer && t == 0) fs.bad[(size_t)((k + 1u) & 1u) * fs.bad_stride + pw] = 0u;
    __syncthreads();
    return true;
  }
  if (t < CELLS) {
    s_pwm[t] = sum;
    s_old[t] = old;
  }
  __syncthreads();
  const float change = finalize_rows<W>(s_pwm, s_old, t);
  const int it = (int)k - 1;
  const bool running = !(change <= threshold || it >= max_it);
  if (writer) {
    float* next = (k & 1u) ? fs.pwm0 : fs.pwm1;  // PWM_(k-1)
    if (t < CELLS) next[(size_t)pw * CELLS + t] = s_pwm[t];
    if (t == 0) {
      fs.state[2 * pw] = it;
      fs.state[2 * pw + 1] = running ? 1 : 0;
      fs.change[pw] = change;
      fs.run[(size_t)((k - 1u) & 1u) * fs.run_stride + pw] = running ? 1u : 0u;
      fs.bad[(size_t)((k + 1u) & 1u) * fs.bad_stride + pw] = 0u;
    }
  }
  return running;
}

// The weights of a span, as em_weights_kernel computes them, and on the way the span's block sums.  A workgroup per span:
// thread t = digits 0..3 of x (lane = digits 0..2: a wave stores 64 consecutive floats), 64 x per thread over digits 4..6.
// The product over the PWM columns in the reference's order ((1*pwm[0][x0])*pwm[1][x1])... (src/peng.cpp:180-197): the
// factors of digits 0..3 once per thread, digit 4 once per 4 x, ...; the digits above the span, equal for all its x, are
// still multiplied last, x by x -- float products do not regroup.
// HEAD (em_serial_scan = 4): the kernel starts with the previous iteration's finalize step (fused_head: every workgroup
// of a PWM repeats it from the cell sums the chains stored; `fs`, `k`, `threshold`, `max_it`) instead of reading the PWM
// and its flags the last chain of the previous launch left -- the chains then need no arrival protocol.
template <int W, bool HEAD>
__global__ __launch_bounds__(256) void em_weights_span_kernel(const float* __restrict__ pwms, const int32_t* __restrict__ state,
                                                              const uint32_t* __restrict__ counts, const float* __restrict__ bg,
                                                              float saturation, float* __restrict__ wbuf, uint32_t* __restrict__ bad,
                                                              float* __restrict__ sums, const uint32_t* __restrict__ bg_range,
                                                              FusedState fs, uint32_t k, float threshold, int max_it) {
  using G = BlockGeo<W>;
  PENGK_WG_TRACE_BEGIN(1);
  const uint32_t pw = blockIdx.y, sp = blockIdx.x;
  __shared__ __attribute__((aligned(16))) float s_pwm[W * 4], s_old[W * 4];
  __shared__ float part[4][28];
  __shared__ uint32_t s_lean;
  const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  uint32_t bg_lo = bg_range[0], bg_hi = bg_range[1];
  if constexpr (HEAD) {
    if (!fused_head<W>(fs, pw, k, threshold, max_it, sp == 0u, s_pwm, s_old, t)) return;
    bad = fs.bad + (size_t)(k & 1u) * fs.bad_stride;
  } else {
    // (the "still running" flag, the PWM and the background's range are asked for TOGETHER -- one memory round trip in front
    // of the span instead of three, in a kernel of ~20 us; the PWM is used before the flag is looked at for that)
    int32_t running = state[2 * pw + 1];
    float mine = pwms[(size_t)pw * W * 4 + (t < W * 4 ? t : 0u)];
    // (one place where all four are needed, in front of the branch: left to itself the compiler asks for the flag, waits,
    // branches, asks for the next ...)
    asm volatile("" : "+s"(running), "+v"(mine), "+s"(bg_lo), "+s"(bg_hi));
    if (t < W * 4) s_pwm[t] = mine;
    if (running == 0) return;
    __syncthreads();
  }
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
__global__ __launch_bounds__(64) void em_block_predict_kernel(const int32_t* __restrict__ state, const uint32_t* __restrict__ bad,
                                                              const float* __restrict__ sums, seqsum::BlockRecord* __restrict__ rec,
                                                              uint32_t skew) {
  using G = BlockGeo<W>;
  const uint32_t pw = blockIdx.y, cell = blockIdx.x, lane = threadIdx.x;
  if (state[2 * pw + 1] == 0 || bad[pw]) return;
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

// A span in LDS, read by every cell that has a block in it: 256 rows of 64 floats; the 16-byte slot k of row R lies at
// slot k ^ g(R), g(R) = (R ^ R >> 3 ^ R >> 4) & 15.  With that, each of the reads below -- a lane fetching four
// consecutive terms of ITS row of 64 terms, for every way a cell's terms lie in the span -- puts the sixteen lanes that
// ds_read_b128 serves together on sixteen different slots (checked for all positions by enumeration; the four cells of
// position 0, every fourth float, read single dwords and pay 4-way conflicts).
struct SpanLds {
  static __device__ __forceinline__ uint32_t g(uint32_t R) { return (R ^ (R >> 3) ^ (R >> 4)) & 15u; }
  static __device__ __forceinline__ uint32_t slot_of(uint32_t R, uint32_t k) { return R * 16u + (k ^ g(R)); }
};

// The terms of lane l's row (terms 64 l .. 64 l + 63 of the block) of task (p, j) of a span, from LDS.
//   p >= 6: block = quarter j of the span (p = 6: the cell (6, j)): x_local = 4096 j + 64 l + i
//   p = 3, 4, 5: digit p = j lies above the low six bits: whole rows, row index = l with j inserted at bit 2 p - 6
//   p = 1, 2: rows 4 l .. 4 l + 3, a quarter of each;  p = 0: every fourth float of those rows
template <int W>
__device__ __forceinline__ void span_row(const float* span, uint32_t p, uint32_t j, uint32_t l, seqsum::Row& row) {
  typedef seqsum::f4 f4;
  const f4* s4 = reinterpret_cast<const f4*>(span);
  if (p >= 3u) {
    uint32_t R;
    if (p >= 6u) {
      R = 64u * j + l;
    } else {
      const uint32_t sh = 2u * p - 6u;
      R = ((l >> sh) << (sh + 2u)) | (j << sh) | (l & ((1u << sh) - 1u));
    }
    const uint32_t base = R * 16u + SpanLds::g(R);
#pragma unroll
    for (uint32_t k = 0; k < 16u; ++k) row.q[k] = s4[base ^ k];
  } else if (p != 0u) {
    // g(4 l + c) = g(4 l) ^ c: slot (s ^ g) of row 4 l + c lies at (64 l + g(4 l)) ^ (16 c + (s ^ c))
    const uint32_t base = 64u * l + SpanLds::g(4u * l);
    if (p == 2u) {
#pragma unroll
      for (uint32_t k = 0; k < 16u; ++k) row.q[k] = s4[base ^ (16u * (k >> 2) + (((k & 3u) + 4u * j) ^ (k >> 2)))];
    } else {
#pragma unroll
      for (uint32_t k = 0; k < 16u; ++k) row.q[k] = s4[base ^ (16u * (k >> 2) + ((4u * (k & 3u) + j) ^ (k >> 2)))];
    }
  } else {
    const uint32_t base = 4u * (64u * l + SpanLds::g(4u * l)) + j;  // (in floats; j < 4 stays below the slot bits)
#pragma unroll
    for (uint32_t k = 0; k < 16u; ++k) {
      // terms 4 k .. 4 k + 3: row 4 l + (k >> 2), slots 4 (k & 3) .. + 3, component j
      const uint32_t c = k >> 2;
      row.q[k].x = span[base ^ (4u * (16u * c + ((4u * (k & 3u) + 0u) ^ c)))];
      row.q[k].y = span[base ^ (4u * (16u * c + ((4u * (k & 3u) + 1u) ^ c)))];
      row.q[k].z = span[base ^ (4u * (16u * c + ((4u * (k & 3u) + 2u) ^ c)))];
      row.q[k].w = span[base ^ (4u * (16u * c + ((4u * (k & 3u) + 3u) ^ c)))];
    }
  }
  asm volatile("" : "+v"(row.q[0]), "+v"(row.q[1]), "+v"(row.q[2]), "+v"(row.q[3]), "+v"(row.q[4]), "+v"(row.q[5]), "+v"(row.q[6]),
               "+v"(row.q[7]), "+v"(row.q[8]), "+v"(row.q[9]), "+v"(row.q[10]), "+v"(row.q[11]), "+v"(row.q[12]), "+v"(row.q[13]),
               "+v"(row.q[14]), "+v"(row.q[15]));
}

// One workgroup per span: the span's 64 KiB are read ONCE (not once per position) into LDS, and the eight waves share
// the 4 W blocks that lie in it -- each block with a predicted binade gets its two increments.
constexpr uint32_t SPAN_EVAL_WAVES = 8;
template <int W>
__global__ __launch_bounds__(64 * SPAN_EVAL_WAVES) void em_span_eval_kernel(const int32_t* __restrict__ state, const float* __restrict__ wbuf,
                                                                            seqsum::BlockRecord* __restrict__ rec,
                                                                            const uint32_t* __restrict__ bad, uint32_t n_pwm,
                                                                            const float* __restrict__ sums, uint32_t skew,
                                                                            uint32_t extra_wgs, const uint32_t* __restrict__ run_now) {
  using G = BlockGeo<W>;
  // (run_now: the PWMs' "still running" flags where the weights kernel keeps them itself (em_serial_scan = 4); else state[])
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
    // (em_chain_kernel: 37 -> 30 us per iteration for 16 PWMs).  Four waves per workgroup (a block staged per wave in a
    // quarter of the span buffer), a cell each; the record says SUM_BEHIND and carries the sum.
    const uint32_t xi = lin, xslot = xi >> 3;
    constexpr uint32_t XW = (G::CELLS + 3u) / 4u;  // workgroups per PWM
    const uint32_t pw = (xi & 7u) + 8u * (xslot / XW), cell = 4u * (xslot % XW) + wave;
    if (pw >= n_pwm || wave >= 4u || cell >= G::CELLS || (run_now ? run_now[pw] == 0u : state[2 * pw + 1] == 0) || bad[pw]) return;
    seqsum::lds_float* buf = (seqsum::lds_float*)span + wave * seqsum::BLOCK;
    const float* w = wbuf + (size_t)pw * G::NP;
    seqsum::Row mine;
    if ((cell >> 2) == 0u) {
      EmTerms0<W> src0{w, cell & 3u};
      src0.bind_stage(lane);
      src0.stage(0u, lane, buf);
    } else {
      EmTerms<W> src{w, cell >> 2, cell & 3u};
      src.bind_stage(lane);
      src.stage(0u, lane, buf);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    mine.read_staged(buf, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    seqsum::Stats st;
    const float s0 = seqsum::fold_block(mine, lane, 0.0f, st);
    if (lane == 0) {
      seqsum::BlockRecord out;
      out.e = seqsum::SUM_BEHIND;
      out.d0 = s0;
      out.d1 = 0.0f;
      out.pad = 0u;
      rec[((size_t)pw * G::CELLS + cell) * G::NBLK] = out;
    }
    PENGK_WG_TRACE_END(1, lin);
    return;
  }
  const uint32_t sl = lin - extra_wgs, slot = sl >> 3;
  const uint32_t pw = (sl & 7u) + 8u * (slot / G::SPANS), sp = slot % G::SPANS;
  if (pw >= n_pwm) return;
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
  const int32_t running = run_now ? (int32_t)run_now[pw] : state[2 * pw + 1];
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
    if (block_of(task) == 0u) continue;  // (folded from zero by the workgroups behind the spans)
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

template <int W>
__global__ __launch_bounds__(64) void em_chain_kernel(int32_t* __restrict__ state, const float* __restrict__ wbuf,
                                                      const seqsum::BlockRecord* __restrict__ rec, double* __restrict__ partials,
                                                      uint32_t* __restrict__ bad, uint32_t n_pwm, uint32_t* __restrict__ done,
                                                      float* __restrict__ pwms, float* __restrict__ change_out, float threshold,
                                                      int max_it, unsigned long long* __restrict__ counters) {
  using G = BlockGeo<W>;
  PENGK_WG_TRACE_BEGIN(3);
  const uint32_t lin = blockIdx.x + gridDim.x * blockIdx.y, slot = lin >> 3;
  const uint32_t cell = slot % G::CELLS, pw = (lin & 7u) + 8u * (slot / G::CELLS);
  if (pw >= n_pwm) return;
  __shared__ __attribute__((aligned(16))) float lds[seqsum::WALK_LDS_FLOATS];
  seqsum::WalkCounts wc;
  const uint32_t lane = threadIdx.x;
  const seqsum::BlockRecord* r = rec + ((size_t)pw * G::CELLS + cell) * G::NBLK;
  // (the PWM's two flags and the chain's first 64 records are asked for together: one memory round trip at the head of
  // every chain -- the kernel ends with its longest one -- instead of three)
  int32_t running = state[2 * pw + 1];
  uint32_t flagged = bad[pw];
  uint4 first = reinterpret_cast<const uint4*>(r)[lane];
  // (one place where all of them are needed, in front of the branch: left to itself the compiler asks for a flag, waits,
  // branches, asks for the next ...)
  asm volatile("" : "+s"(running), "+s"(flagged), "+v"(first.x), "+v"(first.y), "+v"(first.z), "+v"(first.w));
  if (running == 0) return;
  float s = 0.0f;
  const float* w = wbuf + (size_t)pw * G::NP;  // (no second copy in position 0's order: few blocks are read here)
  if (flagged) {  // (a flagged PWM: summed by finalize_pwm's plain loop)
  } else if ((cell >> 2) == 0u) {
    EmTerms0<W> src0{w, cell & 3u};
    src0.bind_stage(lane);
    s = seqsum::walk_chain(src0, r, first, G::NBLK, (seqsum::lds_float*)lds, lane, wc);
  } else {
    EmTerms<W> src{w, cell >> 2, cell & 3u};
    src.bind_stage(lane);
    s = seqsum::walk_chain(src, r, first, G::NBLK, (seqsum::lds_float*)lds, lane, wc);
  }
  // the PWM's last cell to arrive does what em_finalize_kernel does (one launch less per iteration)
  // No fences (a device-scope release writes the XCD's whole L2 back): the sum is stored by a device-scope atomic, which
  // is performed where all XCDs see it, the counter is bumped once that store has returned, and finalize_pwm reads the
  // sums with device-scope loads.
  uint32_t arrived = 0;
  if (lane == 0) {
    // what this chain met (pengk_get_info "em_fetched_blocks" ...): four relaxed adds per chain, nobody waits for them
    if (wc.fetched) __hip_atomic_fetch_add(counters + 0, (unsigned long long)wc.fetched, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wc.mispredicted) __hip_atomic_fetch_add(counters + 1, (unsigned long long)wc.mispredicted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wc.restaged) __hip_atomic_fetch_add(counters + 2, (unsigned long long)wc.restaged, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wc.restaged_waits) __hip_atomic_fetch_add(counters + 3, (unsigned long long)wc.restaged_waits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    (void)__hip_atomic_exchange(reinterpret_cast<unsigned long long*>(partials + (size_t)pw * (W * 4) + cell),
                                (unsigned long long)__double_as_longlong((double)s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    arrived = __hip_atomic_fetch_add(&done[pw], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  arrived = (uint32_t)__builtin_amdgcn_readfirstlane((int)arrived);
  PENGK_WG_TRACE_END(wc.fetched > 255u ? 255u : wc.fetched, lin);  // (kind = blocks taken the long way)
  if (arrived != G::CELLS - 1u) return;
  // The one finalizing wave of a PWM: everything it reads below was written by other workgroups of THIS launch before
  // their fetch_add on done[pw] (the exchange on `partials` returned first).  The acquire fence makes that order explicit
  // for the compiler and the cache (one L2 invalidate per PWM, no write-back -- a release on the 4 W producers would write
  // every XCD's L2 back and tripled the kernel).  What relies on the kernel boundary instead: the plain stores to
  // done[pw], bad[pw], state[], pwms[] and change_out[] below are read by the NEXT launch on this stream only.
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("" ::: "memory");
  if (lane == 0) done[pw] = 0u;
  finalize_pwm<W, 16>((int)pw, pwms, state, change_out, partials, threshold, max_it, bad, wbuf, G::NP, lds);
  PENGK_WG_TRACE_END(254, lin);  // (the PWM's last chain, with the finalize step)
}

// ---- weights, block sums, estimates and block evaluation as ONE kernel (em_serial_scan = 3; W = 10, 12) -----------------
// An iteration of the blocks-ahead scheme above was three dependent launches -- weights (table + block sums), block
// evaluation (the table read back, span by span), chains -- with the finalize step at the end of the last one, behind an
// arrival counter.  Here it is two:
//   em_span_fused_kernel   a workgroup per span: the PREVIOUS iteration's finalize step at its head (every workgroup of a
//                          PWM repeats the 4 W divisions from the cell sums the chains left -- no arrival protocol, no
//                          launch), the span's weights computed into LDS (and stored once, for the chains' fetches),
//                          the span's block sums published, the estimates of the sums in front of its blocks from a
//                          LOOK-BACK over the earlier spans of the PWM, the blocks evaluated from LDS;
//   em_chain_store_kernel  one wave per cell walks the records (seqsum::walk_chain) and stores the cell's sum.
// The table is written once and read only where a chain takes a block the long way (W = 12: 64 MiB per PWM and iteration
// instead of 64 written + 64 read).
//
// The look-back.  Span sp publishes A[sp][cell] = what its weights add to each of the 4 W cells, as 64-bit words {epoch of
// this launch, float}: one relaxed device-scope store per cell, data and "ready" in one word, no fence.  A workgroup adds
// up the words of the earlier spans of its chunk of 64 -- eight per wave, all requested at once -- and the chunk totals
// T[c] of the earlier chunks, which the last span of every chunk publishes the same way.  It only ever waits for
// workgroups with a SMALLER linear index, and the wait is BOUNDED: when the deadline (LOOKBACK_TICKS of the 100 MHz
// clock) passes, the workgroup marks its blocks "no binade" and goes on -- the chain then folds them term by term, which
// costs time and never the result (seqsum.h: exactness does not rest on the estimates).  So neither an unexpected
// dispatch order nor a lost workgroup can hang the launch.  Test hook em_test_lookback = n: every n-th workgroup acts as
// if its deadline had passed.
// per-call counters of what the chains met (pengk_get_info "em_*"): fetched, mispredicted, restaged, restaged_waits
// (seqsum::WalkCounts), blocks passed by their row records, blocks whose row records did not hold (seqsum::Walk2Counts);
// behind them in the same allocation: the background table's {min, max}
constexpr int EM_COUNTERS = 6;
struct FusedGeo {
  static constexpr uint32_t THREADS = 512, WAVES = 8, CHUNK = 64;
  static constexpr unsigned long long LOOKBACK_TICKS = 50000ull;  // 500 us
  static constexpr float ROW_MARGIN = 1.0f / 2048.0f;  // what an estimate is trusted to when it names a ROW's binade (seqsum.h, row_record)
};
// bytes of row records (seqsum.h, "rows ahead of their chain") a cell's blocks may leave per iteration: 2 KiB + 256 B per
// raw row and block; a block that finds no room is folded from the table
template <int W>
struct RowArea {
  static constexpr uint32_t BYTES = W <= 10 ? 64u * 1024u : 256u * 1024u;
};
template <int W>
struct LookGeo {
  using G = BlockGeo<W>;
  static constexpr uint32_t CHUNKS = (G::SPANS + FusedGeo::CHUNK - 1u) / FusedGeo::CHUNK;
  static_assert(CHUNKS <= 64u, "two levels: the earlier spans of a chunk, the earlier chunks");
  static constexpr size_t WORDS_PER_PWM = (size_t)(G::SPANS + CHUNKS) * G::CELLS;  // A[span][cell] | T[chunk][cell]
};
__device__ __forceinline__ unsigned long long look_word(uint32_t epoch, float v) {
  uint32_t b = __float_as_uint(v);
  if (b > 0x7F800000u) b = 0x7F800000u;  // (a NaN or a negative sum -- degenerate weights -- travels as +inf: no binade)
  return ((unsigned long long)epoch << 32) | b;
}
__device__ __forceinline__ unsigned long long look_load(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One weight, the reference's operations (src/peng.cpp:124-125, 180-197): x = the pattern, pr over the PWM's columns in
// position order.
template <int W, bool LEAN>
__device__ __forceinline__ float weight_of(const float* s_pwm, uint32_t x, uint32_t cnt, float b, float saturation) {
  float pr = 1.0f;
#pragma unroll
  for (int p = 0; p < W; ++p) pr = pr * s_pwm[p * 4 + ((x >> (2 * p)) & 3u)];
  const float odds = em_div<LEAN>(pr, b);
  return em_div<LEAN>((float)cnt * saturation, 1 + em_div<LEAN>(saturation, odds));
}

template <int W>
__global__ __launch_bounds__(FusedGeo::THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void em_span_fused_kernel(FusedState fs, uint32_t k, float threshold, int max_it,
                                                                          const uint32_t* __restrict__ counts,
                                                                          const float* __restrict__ bg, float saturation,
                                                                          float* __restrict__ wbuf, seqsum::BlockRecord* __restrict__ rec,
                                                                          unsigned long long* __restrict__ look, uint32_t epoch,
                                                                          const uint32_t* __restrict__ bg_range, uint32_t skew,
                                                                          uint32_t test_lookback, uint32_t extra_wgs,
                                                                          char* __restrict__ row_area, uint32_t* __restrict__ row_cursor) {
  using G = BlockGeo<W>;
  using LG = LookGeo<W>;
  constexpr uint32_t CELLS = G::CELLS, WAVES = FusedGeo::WAVES;
  PENGK_WG_TRACE_BEGIN(2);
  const uint32_t lin = blockIdx.x + gridDim.x * blockIdx.y;
  __shared__ __attribute__((aligned(16))) float span[16384];
  __shared__ __attribute__((aligned(16))) float s_pwm[CELLS], s_old[CELLS];
  __shared__ float s_part[WAVES][28];
  __shared__ float s_cell[CELLS];        // what this span adds to each cell
  __shared__ float s_look[2][WAVES][CELLS];  // the waves' shares of the look-back: [0] earlier spans of the chunk, [1] earlier chunks
  __shared__ uint32_t s_ok[WAVES];
  __shared__ uint32_t s_lean;
  const uint32_t t = threadIdx.x, lane = t & 63u;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(t >> 6));
  uint32_t* bad_now = fs.bad + (size_t)(k & 1u) * fs.bad_stride;
  const uint32_t bg_lo = bg_range[0], bg_hi = bg_range[1];

  if (lin < extra_wgs) {
    // The workgroups IN FRONT of the spans': BLOCK 0 of every cell, whose start is known exactly (zero) and which is the
    // dearest block of a chain -- the sum climbs through some twenty binades in it.  Four cells per workgroup, two waves
    // each: the block's 4096 weights are computed here a second time (a sixth of the table's; nobody to wait for), row by
    // row -- at step i lane l takes term 64 i + l, so that the loads of a step are as contiguous as the cell allows -- into
    // the layout Row::read_staged reads; the even wave of the pair then folds the block from zero.
    const uint32_t xslot = lin >> 3;
    constexpr uint32_t XW = (CELLS + 3u) / 4u;  // workgroups per PWM
    const uint32_t pw = (lin & 7u) + 8u * (xslot / XW), cell = 4u * (xslot % XW) + (wave >> 1);
    if (pw >= fs.n) return;
    if (!fused_head<W>(fs, pw, k, threshold, max_it, false, s_pwm, s_old, t)) return;
    if (t == 0) s_lean = lean_ranges_ok<W>(s_pwm, bg_lo, bg_hi, saturation) ? 1u : 0u;
    __syncthreads();
    const bool lean = s_lean != 0u, live = cell < CELLS;
    const uint32_t p = cell >> 2, a = cell & 3u;
    float* buf = span + (wave >> 1) * seqsum::BLOCK;
    uint32_t worst = 0u;
    auto fill = [&](auto lean_tag) {
      constexpr bool LEAN = decltype(lean_tag)::value;
#pragma unroll 4
      for (uint32_t i = 32u * (wave & 1u); i < 32u * (wave & 1u) + 32u; ++i) {
        const uint32_t c = 64u * i + lane;  // term c of the cell: x = [c's digits p.. | a | c's digits 0..p-1]
        const uint32_t x = ((c >> (2u * p)) << (2u * p + 2u)) | (a << (2u * p)) | (c & ((1u << (2u * p)) - 1u));
        const float v = weight_of<W, LEAN>(s_pwm, x, counts[x], bg[x], saturation);
        worst = max(worst, __float_as_uint(v));
        buf[(16u * i + ((lane >> 2) ^ (i & 15u))) * 4u + (lane & 3u)] = v;  // (row i, Row::read_staged's layout)
      }
    };
    if (live) {
      if (lean) fill(std::true_type{});
      else fill(std::false_type{});
    }
    if (worst > 0x7F7FFFFFu) bad_now[pw] = 1u;
    __syncthreads();
    if (!live || (wave & 1u)) return;
    seqsum::Row mine;
    mine.read_staged((const seqsum::lds_float*)buf, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    seqsum::Stats st;
    // (a flagged PWM: its chains sum the table by the plain loop and look at no record; fold_block's loop ends whatever
    // the terms are -- `first` grows every turn)
    const float s0 = seqsum::fold_block(mine, lane, 0.0f, st);
    if (lane == 0) {
      seqsum::BlockRecord out;
      out.e = seqsum::SUM_BEHIND;
      out.d0 = s0;
      out.d1 = 0.0f;
      out.pad = 0u;
      rec[((size_t)pw * CELLS + cell) * G::NBLK] = out;
    }
    PENGK_WG_TRACE_END(1, lin);
    return;
  }

  const uint32_t sl = lin - extra_wgs, slot = sl >> 3;
  const uint32_t pw = (sl & 7u) + 8u * (slot / G::SPANS), sp = slot % G::SPANS;
  if (pw >= fs.n) return;
  const uint32_t* cnt = counts + (size_t)sp * 16384u;
  const float* bgs = bg + (size_t)sp * 16384u;
  if (!fused_head<W>(fs, pw, k, threshold, max_it, sp == 0u, s_pwm, s_old, t)) return;
  if (t == 0) s_lean = lean_ranges_ok<W>(s_pwm, bg_lo, bg_hi, saturation) ? 1u : 0u;  // (workgroup-uniform: one PWM, one table)
  __syncthreads();
  const bool lean = s_lean != 0u;
  PENGK_WG_TRACE_END(4, lin);  // (the head is done)

  // ---- the span's weights: thread = digits 0..3 of x (t & 255) and the upper half of digit 6 (t >> 8); 32 x per thread
  // over digits 4, 5 and the lower half of digit 6.  The product over the PWM's columns in the reference's order.
  unsigned long long* my_look = look + (size_t)pw * LG::WORDS_PER_PWM;
  // (look-back, below: wave w adds the earlier spans j = w (mod 8) of its chunk and the earlier chunks c = w (mod 8); the
  // words are ASKED FOR here, behind the weights and in front of the reductions and the barrier -- the earlier spans
  // started earlier, most of their words are there by now, and the round trip is hidden)
  const uint32_t chunk = sp / FusedGeo::CHUNK, c0 = chunk * FusedGeo::CHUNK, n0 = sp - c0;
  constexpr uint32_t PER = FusedGeo::CHUNK / WAVES;              // spans per wave
  constexpr uint32_t PERC = (LG::CHUNKS + WAVES - 1u) / WAVES;   // chunk totals per wave
  const unsigned long long* src[PER + PERC];
  unsigned long long v[PER + PERC];
  bool need[PER + PERC];
  {
    float* out_t = wbuf + (size_t)pw * G::NP + (size_t)sp * 16384u + (t & 255u);
    const uint32_t* cnt_t = cnt + (t & 255u);
    const float* bg_t = bgs + (t & 255u);
    const uint32_t tl = t & 255u, h = t >> 8, w3 = (t >> 6) & 3u;
    float p3 = 1.0f;
#pragma unroll
    for (int p = 0; p < 4; ++p) p3 = p3 * s_pwm[p * 4 + ((tl >> (2 * p)) & 3u)];
    float hi[W - 7];  // the span's own digits 7 .. W-1 (wave-uniform)
#pragma unroll
    for (int p = 7; p < W; ++p) hi[p - 7] = s_pwm[p * 4 + ((sp >> (2 * (p - 7))) & 3u)];
    float f4_[4], f5_[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      f4_[a] = s_pwm[16 + a];
      f5_[a] = s_pwm[20 + a];
    }
    float c4[4] = {0, 0, 0, 0}, c5[4] = {0, 0, 0, 0}, c6[2] = {0, 0};
    uint32_t worst = 0u;  // the largest bit pattern among the weights: above +inf's = negative or NaN
    // LDS: float xl of the span lies at 4 (R 16 + (k ^ g(R))) + (xl & 3), R = xl >> 6, k = (xl >> 2) & 15 (SpanLds).  With
    // xl = tl + 256 (d4 + 4 d5 + 16 d6): R = w3 + 4 d4 + 16 d5 + 64 d6 and g(R) = w3 ^ C(d4, d5) ^ G6(d6), so the byte
    // address is (a6 ^ 16 C(d4, d5)) + 256 (4 d4 + 16 d5) with a6 per thread and d6: one xor per x.
    const uint32_t kslot = (tl >> 2) & 15u, comp = tl & 3u;
    char* span_b = reinterpret_cast<char*>(span);
    auto body = [&](auto lean_tag) {
      constexpr bool LEAN = decltype(lean_tag)::value;
#pragma unroll 1
      for (uint32_t i6 = 0; i6 < 2u; ++i6) {  // (not unrolled: 16 x in flight per turn keep the kernel at 128 registers, two workgroups per CU)
        const uint32_t d6 = 2u * h + i6;
        const float f6 = s_pwm[24 + d6];
        const uint32_t g6 = (((d6 & 1u) << 3) ^ ((d6 & 3u) << 2)) & 15u;
        const uint32_t a6 = 256u * (w3 + 64u * d6) + 4u * comp + 16u * ((kslot ^ w3 ^ g6) & 15u);
        float* out6 = out_t + 4096u * d6;
        const uint32_t* cnt6 = cnt_t + 4096u * d6;
        const float* bg6 = bg_t + 4096u * d6;
        float s6 = 0.0f;
#pragma unroll
        for (uint32_t d5 = 0; d5 < 4u; ++d5) {
          float s5 = 0.0f;
#pragma unroll
          for (uint32_t d4 = 0; d4 < 4u; ++d4) {
            constexpr uint32_t dummy = 0u;
            (void)dummy;
            const uint32_t m2 = d4 + 4u * d5;
            const uint32_t cc = ((d4 << 2) ^ (d4 >> 1) ^ (d5 << 1) ^ d5) & 15u;
            float pr = ((p3 * f4_[d4]) * f5_[d5]) * f6;
#pragma unroll
            for (int p = 7; p < W; ++p) pr = pr * hi[p - 7];
            const float odds = em_div<LEAN>(pr, bg6[256u * m2]);
            const float v = em_div<LEAN>((float)cnt6[256u * m2] * saturation, 1 + em_div<LEAN>(saturation, odds));  // :124-125
            out6[256u * m2] = v;
            *reinterpret_cast<float*>(span_b + ((a6 ^ (16u * cc)) + 256u * (4u * d4 + 16u * d5))) = v;
            worst = max(worst, __float_as_uint(v));
            c4[d4] += v;
            s5 += v;
          }
          c5[d5] += s5;
          s6 += s5;
        }
        c6[0] += i6 ? 0.0f : s6;
        c6[1] += i6 ? s6 : 0.0f;
      }
    };
    if (lean) body(std::true_type{});
    else body(std::false_type{});
    if (worst > 0x7F7FFFFFu) bad_now[pw] = 1u;  // (this PWM's cells are summed by the chain kernel's plain loop)
#pragma unroll
    for (uint32_t i = 0; i < PER; ++i) {
      const uint32_t j = wave + WAVES * i;
      need[i] = j < n0 && lane < CELLS;
      src[i] = my_look + (size_t)(c0 + (j < n0 ? j : 0u)) * CELLS + (lane < CELLS ? lane : 0u);
    }
#pragma unroll
    for (uint32_t i = 0; i < PERC; ++i) {
      const uint32_t c = wave + WAVES * i;
      need[PER + i] = c < chunk && lane < CELLS;
      src[PER + i] = my_look + (size_t)(G::SPANS + (c < chunk ? c : 0u)) * CELLS + (lane < CELLS ? lane : 0u);
    }
#pragma unroll
    for (uint32_t i = 0; i < PER + PERC; ++i) v[i] = need[i] ? look_load(src[i]) : 0ull;
    // per wave: whole-wave sums by digit 4, 5, 6; the total by digit 0, 1, 2 (lane bits 0-1, 2-3, 4-5); the total (digit 3)
    const float tot = c6[0] + c6[1];
    auto all = [](float v) {
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
      return v;
    };
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float s4 = all(c4[a]), s5 = all(c5[a]);
      if (lane == 0) {
        s_part[wave][12 + a] = s4;
        s_part[wave][16 + a] = s5;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float s6 = all(c6[i]);
      if (lane == 0) s_part[wave][20 + i] = s6;
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {  // digit d = lane bits 2 d, 2 d + 1: add over the other four bits
      float v = tot;
#pragma unroll
      for (int m = 1; m < 64; m <<= 1)
        if (m != (1 << (2 * d)) && m != (2 << (2 * d))) v += __shfl_xor(v, m, 64);
      if ((lane & ~(3u << (2 * d))) == 0u) s_part[wave][4 * d + (lane >> (2 * d))] = v;
    }
    {
      const float v = all(tot);
      if (lane == 0) s_part[wave][24] = v;
    }
  }
  PENGK_WG_TRACE_END(3, lin);  // (this thread's weights are done)
  __syncthreads();  // the span and the waves' partial sums are in LDS

  // ---- what the span adds to every cell; published for the later spans of the PWM
  if (t < CELLS) {
    const uint32_t p = t >> 2, a = t & 3u;
    auto over_waves = [&](uint32_t at) {
      float v = 0.0f;
#pragma unroll
      for (uint32_t w = 0; w < WAVES; ++w) v += s_part[w][at];
      return v;
    };
    float v;
    if (p <= 2u) v = over_waves(4u * p + a);
    else if (p == 3u) v = s_part[a][24] + s_part[a + 4u][24];  // digit 3 = wave & 3
    else if (p <= 5u) v = over_waves(4u * (p - 1u) + a);        // digits 4, 5 at 12, 16
    else {
      // digit 6 = 2 (wave >> 2) + i; for p >= 7 the span lies in ONE cell of the position, whole
      const uint32_t w0 = 4u * (a >> 1), at = 20u + (a & 1u);
      const float q = (s_part[w0][at] + s_part[w0 + 1u][at]) + (s_part[w0 + 2u][at] + s_part[w0 + 3u][at]);
      if (p == 6u) v = q;
      else v = a == G::high_digit(p, sp) ? over_waves(24u) : 0.0f;
    }
    s_cell[t] = v;
    __hip_atomic_store(my_look + (size_t)sp * CELLS + t, look_word(epoch, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  // ---- look-back: what is not there yet is asked for again until it is, or until the deadline
  {
    bool ok = true;
    const unsigned long long deadline = __builtin_amdgcn_s_memrealtime() + FusedGeo::LOOKBACK_TICKS;
    float acc = 0.0f, acc_spans = 0.0f;
#pragma unroll
    for (uint32_t i = 0; i < PER + PERC; ++i) {
      if (i == PER) {
        acc_spans = acc;
        acc = 0.0f;
      }
      // (the wave polls together: it leaves the loop when no lane waits any more, or at the deadline)
      while (__builtin_amdgcn_ballot_w64(need[i] && (uint32_t)(v[i] >> 32) != epoch) != 0ull) {
        if (__builtin_amdgcn_s_memrealtime() > deadline) {
          ok = false;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
        if (need[i] && (uint32_t)(v[i] >> 32) != epoch) v[i] = look_load(src[i]);
      }
      if (need[i] && (uint32_t)(v[i] >> 32) == epoch) acc += __uint_as_float((uint32_t)v[i]);
    }
    ok = __builtin_amdgcn_ballot_w64(!ok) == 0ull;
    if (lane < CELLS) {
      s_look[0][wave][lane] = acc_spans;
      s_look[1][wave][lane] = acc;
    }
    if (lane == 0) s_ok[wave] = ok ? 1u : 0u;
  }
  __syncthreads();
  bool est_ok = true;
#pragma unroll
  for (uint32_t w = 0; w < WAVES; ++w) est_ok &= s_ok[w] != 0u;
  // the last span of a chunk publishes the chunk's total (earlier spans of the chunk + its own)
  if (est_ok && (sp % FusedGeo::CHUNK) == FusedGeo::CHUNK - 1u && t < CELLS) {
    float tot = s_cell[t];
#pragma unroll
    for (uint32_t w = 0; w < WAVES; ++w) tot += s_look[0][w][t];
    __hip_atomic_store(my_look + (size_t)(G::SPANS + sp / FusedGeo::CHUNK) * CELLS + t, look_word(epoch, tot), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  }
  // (test hook: every n-th workgroup acts as if its deadline had passed -- after it has done its duty to the later chunks)
  if (test_lookback != 0u && (lin * 2654435761u >> 20) % test_lookback == 0u) est_ok = false;
  PENGK_WG_TRACE_END(2, lin);  // (the span is in LDS, the estimates are known)

  // ---- the blocks of the span: task (p, j) -> its cell and block, as in em_span_eval_kernel
  seqsum::BlockRecord* cells = rec + (size_t)pw * CELLS * G::NBLK;
  auto cell_of = [&](uint32_t task) { return (task >> 2) <= 6u ? task : 4u * (task >> 2) + G::high_digit(task >> 2, sp); };
  auto block_of = [&](uint32_t task) { return (task >> 2) <= 6u ? sp : G::high_block(task >> 2, sp, task & 3u); };
#pragma unroll 1
  for (uint32_t task = wave; task < CELLS; task += WAVES) {
    const uint32_t p = task >> 2, j = task & 3u;
    const uint32_t cell = cell_of(task), b = block_of(task);
    if (b == 0u && extra_wgs != 0u) continue;  // (folded from zero by the workgroups in front of the spans')
    seqsum::BlockRecord* r = cells + (size_t)cell * G::NBLK + b;
    uint32_t e = seqsum::NO_BINADE;
    float before = 0.0f;
    if (est_ok) {
#pragma unroll
      for (uint32_t w = 0; w < WAVES; ++w) before += s_look[0][w][cell] + s_look[1][w][cell];
      float own = s_cell[cell];
      if (p >= 7u) {  // quarter j of the span: the quarters in front of it belong to the same cell
        for (uint32_t q = 0; q < j; ++q) before += s_cell[24u + q];
        own = s_cell[24u + j];
      }
      e = block_binade((double)before, (double)before + (double)own, skew, cell * G::NBLK + b);
    }
    const bool by_rows = e == seqsum::NO_BINADE && est_ok && row_area != nullptr;
    if (e == seqsum::NO_BINADE && !by_rows) {
      if (lane == 0) r->e = seqsum::NO_BINADE;
      continue;
    }
    seqsum::Row mine;
    span_row<W>(span, p, j, lane, mine);  // (once per task, whichever way the block goes: the six ways a row lies in the span are the bulk of this loop's code)
    if (by_rows) {
      // No binade for the block as a whole -- block 0, where the sum climbs from zero, or a block in which it passes a
      // power of two: its ROWS are evaluated instead, each under the binade its own estimate names and the one above, and
      // the rows that cannot name one leave their terms behind (seqsum.h, "rows ahead of their chain").  Where to: the
      // cell's area, by a cursor (one returning atomic per such block: ~6 per cell and iteration).
      bool done = false;
      {
        const bool wrong = skew != 0u && (((cell * G::NBLK + b) * 64u + lane) * 2654435761u >> 16) % skew == 0u;
        const seqsum::RowClass rc = seqsum::row_classify(mine, lane, before, FusedGeo::ROW_MARGIN, wrong);
        const bool raw = !rc.certain;
        const unsigned long long rawmask = __builtin_amdgcn_ballot_w64(raw);
        const uint32_t n_raw = (uint32_t)__builtin_popcountll(rawmask);
        if (n_raw <= seqsum::MAX_RAW_ROWS) {
          const uint32_t bytes = seqsum::ROW_RECORD_BYTES + 256u * n_raw;
          uint32_t off = 0u;
          // (the place is asked for as soon as the size is known: the atomic's round trip lies behind the rows' evaluation)
          if (lane == 0) off = __hip_atomic_fetch_add(row_cursor + (size_t)pw * CELLS + cell, bytes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          seqsum::RowRecord rr = seqsum::row_record(mine, rc);
          off = (uint32_t)__builtin_amdgcn_readfirstlane((int)off);
          if (off + bytes <= RowArea<W>::BYTES) {
            char* slot = row_area + ((size_t)pw * CELLS + cell) * RowArea<W>::BYTES + off;
            const uint32_t ridx = __builtin_amdgcn_mbcnt_hi((uint32_t)(rawmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)rawmask, 0u));
            if (raw) rr.info = (rr.info & 0xFFu) | (ridx << 8);
            uint4 ha, hb;
            ha.x = rr.e;
            ha.y = __float_as_uint(rr.d0);
            ha.z = __float_as_uint(rr.d1);
            ha.w = rr.info;
            hb.x = __float_as_uint(rr.u0);
            hb.y = __float_as_uint(rr.u1);
            hb.z = 0u;
            hb.w = 0u;
            reinterpret_cast<uint4*>(slot)[2u * lane] = ha;
            reinterpret_cast<uint4*>(slot)[2u * lane + 1u] = hb;
            if (raw) {
              seqsum::f4* dst = reinterpret_cast<seqsum::f4*>(slot + seqsum::ROW_RECORD_BYTES + 256u * ridx);
#pragma unroll
              for (uint32_t q = 0; q < 16u; ++q) dst[q] = mine.q[q];
            }
            if (lane == 0) {
              seqsum::BlockRecord out;
              out.e = seqsum::ROWS;
              out.d0 = __uint_as_float(off);
              out.d1 = __uint_as_float(n_raw);
              out.pad = 0u;
              *r = out;
            }
            done = true;
          }
        }
      }
      if (!done && lane == 0) r->e = seqsum::NO_BINADE;
      continue;
    }
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

// The chains of a launch of em_span_fused_kernel: one wave per cell; the cell's sum is stored for the next launch's head
// (or em_fused_finish_kernel) -- a plain store, the kernel boundary orders it.
// BY_ROWS: blocks may have left row records (seqsum.h, walk_chain_rows: one buffer for the blocks folded from the table and
// one for the records); else walk_chain with its two block buffers.
template <int W, bool BY_ROWS>
__global__ __launch_bounds__(64) void em_chain_store_kernel(const uint32_t* __restrict__ run, const uint32_t* __restrict__ bad,
                                                            const float* __restrict__ wbuf, const seqsum::BlockRecord* __restrict__ rec,
                                                            float* __restrict__ cellsum, uint32_t n_pwm,
                                                            unsigned long long* __restrict__ counters,
                                                            const char* __restrict__ row_area, uint32_t* __restrict__ row_cursor) {
  using G = BlockGeo<W>;
  PENGK_WG_TRACE_BEGIN(3);
  const uint32_t lin = blockIdx.x + gridDim.x * blockIdx.y, slot = lin >> 3;
  const uint32_t cell = slot % G::CELLS, pw = (lin & 7u) + 8u * (slot / G::CELLS);
  if (pw >= n_pwm) return;
  __shared__ __attribute__((aligned(16))) float lds[BY_ROWS ? seqsum::WALK2_LDS_FLOATS : seqsum::WALK_LDS_FLOATS];
  seqsum::Walk2Counts wc;
  seqsum::WalkCounts wc1;
  const char* rows = BY_ROWS ? row_area + ((size_t)pw * G::CELLS + cell) * RowArea<W>::BYTES : nullptr;
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
    if constexpr (BY_ROWS) s = seqsum::walk_chain_rows(src0, r, first, G::NBLK, rows, (seqsum::lds_float*)lds, lane, wc);
    else s = seqsum::walk_chain(src0, r, first, G::NBLK, (seqsum::lds_float*)lds, lane, wc1);
  } else {
    EmTerms<W> src{w, cell >> 2, cell & 3u};
    src.bind_stage(lane);
    if constexpr (BY_ROWS) s = seqsum::walk_chain_rows(src, r, first, G::NBLK, rows, (seqsum::lds_float*)lds, lane, wc);
    else s = seqsum::walk_chain(src, r, first, G::NBLK, (seqsum::lds_float*)lds, lane, wc1);
  }
  if constexpr (!BY_ROWS) {
    wc.fetched = wc1.fetched;
    wc.mispredicted = wc1.mispredicted;
  }
  if (lane == 0) {
    cellsum[(size_t)pw * G::CELLS + cell] = s;
    if (BY_ROWS) row_cursor[(size_t)pw * G::CELLS + cell] = 0u;  // (the next launch's blocks start over in the cell's area)
    if (wc1.restaged) __hip_atomic_fetch_add(counters + 2, (unsigned long long)wc1.restaged, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wc1.restaged_waits) __hip_atomic_fetch_add(counters + 3, (unsigned long long)wc1.restaged_waits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wc.fetched) __hip_atomic_fetch_add(counters + 0, (unsigned long long)wc.fetched, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wc.mispredicted) __hip_atomic_fetch_add(counters + 1, (unsigned long long)wc.mispredicted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wc.rows) __hip_atomic_fetch_add(counters + 4, (unsigned long long)wc.rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wc.rows_failed) __hip_atomic_fetch_add(counters + 5, (unsigned long long)wc.rows_failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  PENGK_WG_TRACE_END(wc.fetched > 255u ? 255u : wc.fetched, lin);
}

// Once per call and batch, behind the last launch's chains: F_K for the PWMs that are still running (K = launches = the
// iteration limit: they stop here), and every PWM's final matrix into the caller's array (PWM_j lives there for even j).
template <int W>
__global__ __launch_bounds__(64) void em_fused_finish_kernel(FusedState fs, uint32_t K, float threshold, int max_it) {
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
__global__ __launch_bounds__(256) void em_fused_setup_kernel(uint32_t n, int W, float threshold, int max_it, int32_t* __restrict__ state,
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

__global__ void em_init_kernel(int n, int W, float threshold, int max_it, int32_t* __restrict__ state,
                               float* __restrict__ change) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float c0 = (float)W;  // `float change = pattern_length` (src/peng.cpp:101)
  state[2 * i] = 0;
  state[2 * i + 1] = !(c0 <= threshold || 0 >= max_it);
  change[i] = c0;
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

// The serial mode with the blocks evaluated ahead of their chain (em_serial_scan = 2, W >= 10), on SEVERAL streams: the
// PWMs go round in batches, and the batches take turns on the context's stream and up to three more.  A batch's iteration
// is weights -> block evaluation -> chains, the first two bound by arithmetic and the last by one wave per cell waiting
// for its next block; with several batches in flight the waits of one are filled by the others.  (Left to themselves the
// lanes fall into step -- chains beside chains, weights beside weights: 0.83 ms for 16 PWMs x 10 iterations at W = 10
// against 0.92 on one stream.  Holding the second lane back until the first one's first weights / evaluation / chain
// kernel has run, so that chains run beside weights, was measured twice: no gain, behind the chains a loss --
// profiles/r04_em_kernels.log.)
// (PWMs are independent; every batch has its own tables, records and sums.)  `budget` = bytes of weight tables in flight.
// What a pengk_em call in this mode starts from, in ONE launch (four to five memsets took 8-25 us apiece in front of the
// first weights kernel): the chains' counters zero, the background's range {all ones, 0} for em_bg_range_kernel's min /
// max -- or {0, all ones}, a range nothing accepts, with the lean division off --, every lane's flags and arrival
// counters zero.
__global__ __launch_bounds__(256) void em_ahead_setup_kernel(unsigned long long* __restrict__ counters, uint32_t* __restrict__ bg_range,
                                                             uint32_t lean, char* __restrict__ partials, size_t partials_b, size_t flags_at,
                                                             uint32_t flag_words, uint32_t lanes) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  if (t < (uint32_t)EM_COUNTERS) counters[t] = 0ull;
  if (t == 0) {
    bg_range[0] = lean ? 0xFFFFFFFFu : 0u;
    bg_range[1] = lean ? 0u : 0xFFFFFFFFu;
  }
  for (uint32_t l = 0; l < lanes; ++l) {
    uint32_t* f = reinterpret_cast<uint32_t*>(partials + l * partials_b + flags_at);
    for (uint32_t i = t; i < flag_words; i += gridDim.x * 256u) f[i] = 0u;
  }
}
template <int W>
int launch_serial_ahead(pengk_ctx* ctx, int64_t n_pwm, float* d_pwms, float saturation, float threshold, int max_it,
                        const uint32_t* d_counts, const float* d_bg, int32_t* d_state, float* d_change, size_t budget) {
  using G = EmGeo<W, 16>;
  using B = BlockGeo<W>;
  const size_t np = (size_t)1 << (2 * W);
  // lanes: option "em_overlap" (1 = one stream), as many as leave a lane at least eight PWMs (two by default: 16 PWMs x
  // 10 iterations at W = 10 take 0.92 / 0.83 / 0.86 / 0.84 ms on 1 / 2 / 3 / 4 streams, 1000 PWMs 38.5 / 33.8 / 34.5 / 39.5 ms;
  // four lanes of four PWMs, two lanes per half of the XCDs: 0.98 ms, also inside bench.py's step, where the host is 3 ms ahead --
  // twice the kernels, each with its own ramp and tail, is what costs, not the enqueueing)
  int lanes = ctx->em_overlap < 1 ? 1 : ctx->em_overlap > MAX_EM_LANES ? MAX_EM_LANES : ctx->em_overlap;
  while (lanes > 1 && n_pwm < 8 * (int64_t)lanes) --lanes;
  const int64_t fit = std::max<int64_t>(1, (int64_t)(budget / lanes / (np * sizeof(float))));  // tables the budget holds per lane
  int64_t batch = fit;
  if (batch * lanes > n_pwm) {
    // fewer PWMs than the lanes could hold: equal shares, in whole groups of 8 PWMs (one per XCD) -- but never more
    // tables than the budget (17 PWMs on 2 lanes with room for 9 each: 8 + 8 + 1 in three turns, not 2 x 16)
    batch = lanes > 1 ? ((n_pwm + lanes - 1) / lanes + 7) / 8 * 8 : n_pwm;
    if (batch > fit) batch = fit >= 8 ? fit / 8 * 8 : fit;
  }
  if (batch > 65528) batch = 65528;  // gridDim.y
  // per lane: tables | cell sums, flags ("has a weight the scan cannot take"), arrival counters | block sums, records
  const size_t tables_b = (size_t)batch * np * sizeof(float);
  const size_t flags_at = (size_t)batch * G::CELLS * sizeof(double);
  const size_t partials_b = (flags_at + (size_t)2 * batch * sizeof(uint32_t) + 255) / 256 * 256;
  const size_t blocks_b = ((size_t)batch * B::CELLS * B::NBLK * (sizeof(float) + sizeof(seqsum::BlockRecord)) + 255) / 256 * 256;
  int rc = ensure_scratch(ctx, (void**)&ctx->d_em_tables, &ctx->em_tables_bytes, lanes * tables_b);
  if (rc) return rc;
  rc = ensure_scratch(ctx, (void**)&ctx->d_em_partials, &ctx->em_partials_bytes, lanes * partials_b);
  if (rc) return rc;
  rc = ensure_scratch(ctx, &ctx->d_em_blocks, &ctx->em_blocks_bytes, lanes * blocks_b);
  if (rc) return rc;
  if (!ctx->d_em_counters) PENGK_HIP(hipMalloc((void**)&ctx->d_em_counters, (EM_COUNTERS + 1) * sizeof(unsigned long long)));
  // the range of the background table, for the weights kernel's choice of division (lean_ranges_ok)
  uint32_t* bg_range = reinterpret_cast<uint32_t*>(ctx->d_em_counters + EM_COUNTERS);
  {
    const uint32_t flag_words = (uint32_t)(2 * batch);  // (batch <= 65528)
    hipLaunchKernelGGL(em_ahead_setup_kernel, dim3((flag_words + 255u) / 256u), dim3(256), 0, ctx->stream, ctx->d_em_counters, bg_range,
                       ctx->em_lean_div ? 1u : 0u, reinterpret_cast<char*>(ctx->d_em_partials), partials_b, flags_at, flag_words,
                       (uint32_t)lanes);
    if (ctx->em_lean_div)
      hipLaunchKernelGGL(em_bg_range_kernel, dim3(32), dim3(1024), 0, ctx->stream, d_bg, (uint32_t)np, bg_range);  // (np = 4^W: a multiple of 4)
  }
  hipStream_t streams[MAX_EM_LANES];
  streams[0] = ctx->stream;
  for (int l = 1; l < lanes; ++l) {
    if (!ctx->em_streams[l - 1]) PENGK_HIP(hipStreamCreateWithFlags(&ctx->em_streams[l - 1], hipStreamNonBlocking));
    if (!ctx->em_join[l - 1]) PENGK_HIP(hipEventCreateWithFlags(&ctx->em_join[l - 1], hipEventDisableTiming));
    streams[l] = ctx->em_streams[l - 1];
  }
  if (lanes > 1 && !ctx->em_fork) PENGK_HIP(hipEventCreateWithFlags(&ctx->em_fork, hipEventDisableTiming));
  if (lanes > 1) {  // (everything enqueued so far -- the tables' producers, em_init_kernel -- comes first on all of them)
    PENGK_HIP(hipEventRecord(ctx->em_fork, ctx->stream));
    for (int l = 1; l < lanes; ++l) PENGK_HIP(hipStreamWaitEvent(streams[l], ctx->em_fork, 0));
  }
  // Once the lanes are forked they are ALWAYS joined, also when a launch fails half way: the caller reads and frees
  // buffers on ctx->stream, and kernels may still run on the other streams.
  // The batches go to the lanes in turn, and the launches are ENQUEUED in turn as well: iteration 1 of every lane's batch,
  // then iteration 2 ...  Batch by batch -- all iterations of lane 0's, then all of lane 1's -- the second lane got its
  // first kernel only when the host had enqueued the first lane's thirty launches: with 16 PWMs the first lane was two
  // thirds through its ten iterations by then (profiles/r04_em_kernels.log, the timeline).
  const int rc_launch = [&]() -> int {
    for (int64_t round0 = 0; round0 < n_pwm; round0 += batch * lanes) {
      for (int it = 0; it < max_it; ++it) {
        for (int l = 0; l < lanes; ++l) {
          const int64_t first = round0 + (int64_t)l * batch;
          if (first >= n_pwm) break;
          const int64_t nb = n_pwm - first < batch ? n_pwm - first : batch;
          hipStream_t st = streams[l];
          float* tables = reinterpret_cast<float*>(reinterpret_cast<char*>(ctx->d_em_tables) + l * tables_b);
          double* partials = reinterpret_cast<double*>(reinterpret_cast<char*>(ctx->d_em_partials) + l * partials_b);
          uint32_t* bad = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(partials) + flags_at);
          uint32_t* done = bad + batch;
          float* sums = reinterpret_cast<float*>(reinterpret_cast<char*>(ctx->d_em_blocks) + l * blocks_b);
          seqsum::BlockRecord* rec = reinterpret_cast<seqsum::BlockRecord*>(sums + (size_t)batch * B::CELLS * B::NBLK);
          hipLaunchKernelGGL((em_weights_span_kernel<W, false>), dim3(B::SPANS, (unsigned)nb), dim3(256), 0, st, d_pwms + (size_t)first * W * 4,
                             d_state + 2 * first, d_counts, d_bg, saturation, tables, bad, sums, (const uint32_t*)bg_range, FusedState{}, 0u,
                             0.0f, 0);
          if (!B::PREDICT_IN_EVAL)
            hipLaunchKernelGGL((em_block_predict_kernel<W>), dim3(B::CELLS, (unsigned)nb), dim3(64), 0, st, d_state + 2 * first, bad, sums, rec,
                               (uint32_t)ctx->em_test_skew);
          const unsigned groups = (unsigned)((nb + 7) / 8 * 8);  // (PWMs in whole groups of 8, one per XCD)
          const uint64_t extra_wgs = (uint64_t)groups * ((B::CELLS + 3) / 4);  // block 0 of every cell, in front of ...
          const uint64_t wgs = extra_wgs + (uint64_t)groups * B::SPANS;         // ... the spans
          const unsigned gx = 1024u;
          hipLaunchKernelGGL((em_span_eval_kernel<W>), dim3(gx, (unsigned)((wgs + gx - 1) / gx)), dim3(64 * SPAN_EVAL_WAVES), 0, st,
                             d_state + 2 * first, (const float*)tables, rec, bad, (uint32_t)nb, (const float*)sums, (uint32_t)ctx->em_test_skew,
                             (uint32_t)extra_wgs, (const uint32_t*)nullptr);
          hipLaunchKernelGGL((em_chain_kernel<W>), dim3((unsigned)(4 * W), groups), dim3(64), 0, st, d_state + 2 * first, (const float*)tables,
                             (const seqsum::BlockRecord*)rec, partials, bad, (uint32_t)nb, done, d_pwms + (size_t)first * W * 4,
                             d_change + first, threshold, max_it, ctx->d_em_counters);
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

// The serial mode as two launches per iteration (em_serial_scan = 3; W = 10, 12): em_span_fused_kernel + em_chain_store_kernel
// per batch of PWMs, batches taking turns on the lanes as in launch_serial_ahead, one em_fused_finish_kernel per batch.
// `split` (em_serial_scan = 4): three launches per iteration as in launch_serial_ahead -- weights, block evaluation, chains --
// but with the finalize step at the head of the weights kernel and chains that store their sums plainly.
template <int W>
int launch_serial_fused(pengk_ctx* ctx, int64_t n_pwm, float* d_pwms, float saturation, float threshold, int max_it,
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
  const size_t look_b = ((size_t)batch * LG::WORDS_PER_PWM * sizeof(unsigned long long) + 255) / 256 * 256;
  const size_t flags_at = ((size_t)batch * B::CELLS * sizeof(float) + 255) / 256 * 256;
  const size_t cursor_at = (flags_at + (size_t)2 * batch * sizeof(uint32_t) + 255) / 256 * 256;  // (the row records' cursors: cleared with the flags)
  const size_t small_b = (cursor_at + (size_t)batch * B::CELLS * sizeof(uint32_t) + 255) / 256 * 256;
  const size_t rows_b = ctx->em_rows && !split ? (size_t)batch * B::CELLS * RowArea<W>::BYTES : 0;
  const size_t run_at = lanes * small_b;
  const size_t pwm1_at = (run_at + (size_t)2 * n_pwm * sizeof(uint32_t) + 255) / 256 * 256;
  int rc = ensure_scratch(ctx, (void**)&ctx->d_em_tables, &ctx->em_tables_bytes, lanes * tables_b);
  if (rc) return rc;
  rc = ensure_scratch(ctx, (void**)&ctx->d_em_partials, &ctx->em_partials_bytes, pwm1_at + (size_t)n_pwm * B::CELLS * sizeof(float));
  if (rc) return rc;
  rc = ensure_scratch(ctx, &ctx->d_em_blocks, &ctx->em_blocks_bytes, lanes * rec_b);
  if (rc) return rc;
  if (rows_b) {
    rc = ensure_scratch(ctx, &ctx->d_em_rows, &ctx->em_rows_bytes, lanes * rows_b);
    if (rc) return rc;
  }
  {
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
    hipLaunchKernelGGL(em_fused_setup_kernel, dim3((unsigned)((std::max<int64_t>(n_pwm, 1024) + 255) / 256)), dim3(256), 0, ctx->stream,
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
          uint32_t* cursor = reinterpret_cast<uint32_t*>(small + l * small_b + cursor_at);
          char* rows = rows_b ? reinterpret_cast<char*>(ctx->d_em_rows) + l * rows_b : nullptr;
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
            hipLaunchKernelGGL((em_fused_finish_kernel<W>), dim3((unsigned)nb), dim3(64), 0, st, fs, (uint32_t)max_it, threshold, max_it);
            continue;
          }
          const uint32_t k = (uint32_t)it;
          const unsigned groups = (unsigned)((nb + 7) / 8 * 8);  // (PWMs in whole groups of 8, one per XCD)
          const uint32_t* run_now = run + (size_t)((k - 1u) & 1u) * n_pwm + first;  // behind this launch's head
          const uint32_t* bad_now = bad + (size_t)(k & 1u) * batch;
          if (split) {
            hipLaunchKernelGGL((em_weights_span_kernel<W, true>), dim3(B::SPANS, (unsigned)nb), dim3(256), 0, st, (const float*)nullptr,
                               (const int32_t*)nullptr, d_counts, d_bg, saturation, tables, (uint32_t*)nullptr, sums, (const uint32_t*)bg_range, fs,
                               k, threshold, max_it);
            const uint64_t xwgs = (uint64_t)groups * ((B::CELLS + 3) / 4), swgs = xwgs + (uint64_t)groups * B::SPANS;
            hipLaunchKernelGGL((em_span_eval_kernel<W>), dim3(1024u, (unsigned)((swgs + 1023u) / 1024u)), dim3(64 * SPAN_EVAL_WAVES), 0, st,
                               (const int32_t*)nullptr, (const float*)tables, rec, bad_now, (uint32_t)nb, (const float*)sums,
                               (uint32_t)ctx->em_test_skew, (uint32_t)xwgs, run_now);
            hipLaunchKernelGGL((em_chain_store_kernel<W, false>), dim3((unsigned)(4 * W), groups), dim3(64), 0, st, run_now, bad_now,
                               (const float*)tables, (const seqsum::BlockRecord*)rec, cellsum, (uint32_t)nb, ctx->d_em_counters,
                               (const char*)nullptr, cursor);
            continue;
          }
          // block 0 of every cell: by workgroups in front of the spans' (option em_block0 = 1), or left to its chain
          const uint64_t extra_wgs = ctx->em_block0 ? (uint64_t)groups * ((B::CELLS + 3) / 4) : 0;
          const uint64_t wgs = extra_wgs + (uint64_t)groups * B::SPANS;
          const unsigned gx = 1024u;
          const uint32_t epoch = ++ctx->em_epoch;
          // The lanes out of step: lane l's first span kernel waits for lane l - 1's -- started together, the lanes run their
          // span kernels side by side (each at half the chip) and then their chains side by side; one behind the other, a
          // lane's chains -- waves that wait for memory -- run beside the other lane's arithmetic (option em_stagger).
          if (ctx->em_stagger && it == 1 && round0 == 0 && l > 0 && ctx->em_step[l - 1])
            PENGK_HIP(hipStreamWaitEvent(st, ctx->em_step[l - 1], 0));
          hipLaunchKernelGGL((em_span_fused_kernel<W>), dim3(gx, (unsigned)((wgs + gx - 1) / gx)), dim3(FusedGeo::THREADS), 0, st, fs, k,
                             threshold, max_it, d_counts, d_bg, saturation, tables, rec, look, epoch, (const uint32_t*)bg_range,
                             (uint32_t)ctx->em_test_skew, (uint32_t)ctx->em_test_lookback, (uint32_t)extra_wgs, rows, cursor);
          if (ctx->em_stagger && it == 1 && round0 == 0 && l + 1 < lanes) {
            if (!ctx->em_step[l]) PENGK_HIP(hipEventCreateWithFlags(&ctx->em_step[l], hipEventDisableTiming));
            PENGK_HIP(hipEventRecord(ctx->em_step[l], st));
          }
          if (rows)
            hipLaunchKernelGGL((em_chain_store_kernel<W, true>), dim3((unsigned)(4 * W), groups), dim3(64), 0, st, run_now, bad_now,
                               (const float*)tables, (const seqsum::BlockRecord*)rec, cellsum, (uint32_t)nb, ctx->d_em_counters,
                               (const char*)rows, cursor);
          else
            hipLaunchKernelGGL((em_chain_store_kernel<W, false>), dim3((unsigned)(4 * W), groups), dim3(64), 0, st, run_now, bad_now,
                               (const float*)tables, (const seqsum::BlockRecord*)rec, cellsum, (uint32_t)nb, ctx->d_em_counters,
                               (const char*)rows, cursor);
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
template <int W>
int launch_serial(pengk_ctx* ctx, int64_t n_pwm, float* d_pwms, float saturation, float threshold, int max_it,
                  const uint32_t* d_counts, const float* d_bg, int32_t* d_state, float* d_change) {
  using G = EmGeo<W, 16>;  // HI = W - 4, one partial block per PWM
  static_assert(G::NB == 1, "serial mode writes one row of cells per PWM");
  static_assert(W >= 4, "a cell has at least 16 terms");
  hipLaunchKernelGGL(em_init_kernel, dim3((unsigned)((n_pwm + 255) / 256)), dim3(256), 0, ctx->stream, (int)n_pwm, W, threshold,
                     max_it, d_state, d_change);
  PENGK_HIP(hipGetLastError());
  const size_t np = (size_t)1 << (2 * W);
  // Weight tables of one batch of PWMs (4^W floats per PWM, twice that with the scan's permuted copy: 8 MiB at W = 10,
  // 128 MiB at W = 12); a batch is one launch per kernel and iteration.  Option "em_table_budget_mb" (0 = automatic):
  //  * tables of up to 16 MiB per PWM (W <= 10): 192 MiB per batch, so that what the weights kernel writes is still in
  //    the 256 MiB Infinity Cache when the scan reads it W times (1000 PWMs x 10 iterations at W = 10: 48.6 ms; with
  //    1 GiB batches 58 ms, with all PWMs in one batch 79 ms: the scan is bound by its table reads, not by arithmetic;
  //    profiles/r03_em_budget.log);
  //  * larger tables never fit: one batch for the whole call (a quarter of the free memory at most), i.e. one launch
  //    per kernel and iteration -- the scan then streams 12 x 64 MiB per PWM and iteration from HBM at W = 12.
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
  constexpr bool SCAN = (1u << (2 * W - 2)) >= 4u * seqsum::BLOCK;
  const bool scan = SCAN && ctx->em_serial_scan != 0;
  constexpr bool COPY0 = SCAN && ScanCopy0<W>::value;
  // blocks evaluated ahead of the chain (W >= 10): span-major weights with block sums, no second copy of the table
  if constexpr (SCAN && W >= 10) {
    if constexpr (W <= 12) {  // (W = 14: 16384 spans per PWM would take a third look-back level; it keeps the three launches)
      if (scan && ctx->em_serial_scan >= 3)
        return launch_serial_fused<W>(ctx, n_pwm, d_pwms, saturation, threshold, max_it, d_counts, d_bg, d_state, d_change, budget,
                                      ctx->em_serial_scan == 4);
    }
    if (scan && ctx->em_serial_scan >= 2)
      return launch_serial_ahead<W>(ctx, n_pwm, d_pwms, saturation, threshold, max_it, d_counts, d_bg, d_state, d_change, budget);
  }
  const size_t pwm_stride = (scan && COPY0) ? 2 * np : np;  // floats per PWM: the weight table (+ its copy in position 0's order)
  int64_t batch = (int64_t)(budget / (pwm_stride * sizeof(float)));
  if (batch < 1) batch = 1;
  if (batch > n_pwm) batch = n_pwm;
  if (batch > 65528) batch = 65528;  // gridDim.y, in whole groups of 8 PWMs
  int rc = ensure_scratch(ctx, (void**)&ctx->d_em_tables, &ctx->em_tables_bytes, (size_t)batch * pwm_stride * sizeof(float));
  if (rc) return rc;
  // partials: one row of cells per PWM, and behind them one flag per PWM ("has a weight the scan cannot take")
  const size_t flags_at = (size_t)batch * G::CELLS * sizeof(double);
  rc = ensure_scratch(ctx, (void**)&ctx->d_em_partials, &ctx->em_partials_bytes, flags_at + (size_t)batch * sizeof(uint32_t));
  if (rc) return rc;
  uint32_t* bad = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(ctx->d_em_partials) + flags_at);
  PENGK_HIP(hipMemsetAsync(bad, 0, (size_t)batch * sizeof(uint32_t), ctx->stream));
  // cells of at least four blocks are summed by the scan (seqsum.h; flagged PWMs by the finalize kernel's plain loop),
  // the short chains of W <= 6 by the dependent-addition fold
  const unsigned wb = (unsigned)std::min<size_t>((np / 16 + 255) / 256, 1024);  // a thread per 16 x
  for (int64_t first = 0; first < n_pwm; first += batch) {
    const int64_t nb = n_pwm - first < batch ? n_pwm - first : batch;
    for (int it = 0; it < max_it; ++it) {
      if (scan && COPY0)
        hipLaunchKernelGGL((em_weights_kernel<W, true>), dim3(wb, (unsigned)nb), dim3(256), 0, ctx->stream, d_pwms + (size_t)first * W * 4,
                           d_state + 2 * first, d_counts, d_bg, saturation, ctx->d_em_tables, bad);
      else
        hipLaunchKernelGGL((em_weights_kernel<W, false>), dim3(wb, (unsigned)nb), dim3(256), 0, ctx->stream, d_pwms + (size_t)first * W * 4,
                           d_state + 2 * first, d_counts, d_bg, saturation, ctx->d_em_tables, bad);
      bool scanned = false;
      if constexpr (SCAN) {
        if (scan) {
          hipLaunchKernelGGL((em_fold_scan_kernel<W>), dim3((unsigned)(4 * W), (unsigned)((nb + 7) / 8 * 8)), dim3(seqsum::CHAIN_THREADS), 0, ctx->stream,
                             d_state + 2 * first, ctx->d_em_tables, ctx->d_em_partials, bad, (uint32_t)nb);
          scanned = true;
        }
      }
      if (!scanned)
        hipLaunchKernelGGL((em_fold_kernel<W>), dim3((unsigned)W, (unsigned)nb), dim3(192), 0, ctx->stream, d_state + 2 * first,
                           ctx->d_em_tables, ctx->d_em_partials, (uint32_t)pwm_stride);
      hipLaunchKernelGGL((em_finalize_kernel<W, 16>), dim3((unsigned)nb), dim3(64), 0, ctx->stream,
                         d_pwms + (size_t)first * W * 4, d_state + 2 * first, d_change + first, ctx->d_em_partials, threshold, max_it,
                         scanned ? bad : (uint32_t*)nullptr, (const float*)ctx->d_em_tables, (uint32_t)pwm_stride);
    }
    PENGK_HIP(hipGetLastError());
  }
  return PENGK_OK;
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
