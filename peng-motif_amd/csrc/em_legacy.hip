// em_legacy.hip -- K5, serial mode (em_fast = 2): the generations before the blocks-ahead scheme of em.hip.
//
// The reference adds the 4^W weights into each PWM cell one after the other in float32 (src/peng.cpp:121-127); these kernels
// produce that sum bit for bit in the two ways the library did before its blocks were evaluated ahead of their chain:
//   generation 0   em_weights_kernel + em_fold_kernel: all weights of a PWM in parallel, then one workgroup per position
//                  whose adder lanes add their cell's terms strictly in order (the product's path for W = 4, 6);
//   generation 1   em_weights_kernel + em_fold_scan_kernel: one wave per cell evaluates the chain of roundings as a scan,
//                  block after block (seqsum.h, fold_chain; the product's path for W = 8).
// For W >= 10 they are cross-checks: tests/ select them through pengk_test_em_generation and compare with the product.
#include "em_common.h"

namespace pengk {
namespace {

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


template <int W>
int launch_generation(pengk_ctx* ctx, int generation, int64_t n_pwm, float* d_pwms, float saturation, float threshold, int max_it,
                      const uint32_t* d_counts, const float* d_bg, int32_t* d_state, float* d_change, size_t budget) {
  using G = EmGeo<W, 16>;  // HI = W - 4, one partial block per PWM
  static_assert(G::NB == 1, "serial mode writes one row of cells per PWM");
  static_assert(W >= 4, "a cell has at least 16 terms");
  hipLaunchKernelGGL(em_init_kernel, dim3((unsigned)((n_pwm + 255) / 256)), dim3(256), 0, ctx->stream, (int)n_pwm, W, threshold,
                     max_it, d_state, d_change);
  PENGK_HIP(hipGetLastError());
  const size_t np = (size_t)1 << (2 * W);
  constexpr bool SCAN = (1u << (2 * W - 2)) >= 4u * seqsum::BLOCK;
  const bool scan = SCAN && generation != 0;
  constexpr bool COPY0 = SCAN && ScanCopy0<W>::value;
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

}  // namespace

int launch_em_serial_legacy(pengk_ctx* ctx, int W, int generation, int64_t n_pwm, float* d_pwms, float saturation, float threshold,
                            int max_it, const uint32_t* d_counts, const float* d_bg, int32_t* d_state, float* d_change, size_t budget) {
#define PENGK_GEN(WW) \
  case WW: return launch_generation<WW>(ctx, generation, n_pwm, d_pwms, saturation, threshold, max_it, d_counts, d_bg, d_state, d_change, budget)
  switch (W) {
    PENGK_GEN(4);
    PENGK_GEN(6);
    PENGK_GEN(8);
    PENGK_GEN(10);
    PENGK_GEN(12);
    PENGK_GEN(14);
    default: return fail(PENGK_ERR_ARG, "pattern length %d unsupported", W);
  }
#undef PENGK_GEN
}

}  // namespace pengk
