// em_fused.hip -- K5, serial mode, opt-in (em_serial_scan = 3): weights, block sums, estimates and block evaluation of a span
// as ONE kernel.  Bit-exact like the three-launch scheme of em.hip, measured slower at W = 10 and level at W = 12
// (profiles/r05_em_kernels.log); kept as the record of what VERDICT r04 asked for first and as a fourth
// independent path in the differential soaks (tests/tools/em_scan_fuzz.py).
#include "em_serial.h"

// (the workgroup trace of the developer build -DPENGK_WG_TRACE lives in em.hip; this kernel's marks were taken in round 5:
// profiles/r05_em_kernels.log section 3)
#define PENGK_WG_TRACE_BEGIN(k)
#define PENGK_WG_TRACE_END(kind, wg)

namespace pengk {
namespace {
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
__device__ __forceinline__ unsigned long long look_word(uint32_t epoch, float v) {
  uint32_t b = __float_as_uint(v);
  if (b > 0x7F800000u) b = 0x7F800000u;  // (a NaN or a negative sum -- degenerate weights -- travels as +inf: no binade)
  return ((unsigned long long)epoch << 32) | b;
}
__device__ __forceinline__ unsigned long long look_load(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int W>
__global__ __launch_bounds__(FusedGeo::THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void em_span_fused_kernel(FusedState fs, uint32_t k, float threshold, int max_it,
                                                                          const uint32_t* __restrict__ counts,
                                                                          const float* __restrict__ bg, float saturation,
                                                                          float* __restrict__ wbuf, seqsum::BlockRecord* __restrict__ rec,
                                                                          unsigned long long* __restrict__ look, uint32_t epoch,
                                                                          const uint32_t* __restrict__ bg_range, uint32_t skew,
                                                                          uint32_t test_lookback) {
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

  const uint32_t sl = lin, slot = sl >> 3;
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
    if (e == seqsum::NO_BINADE) {  // (block 0, where the sum climbs from zero, and the blocks where it passes a power of two: folded by the chain)
      if (lane == 0) r->e = seqsum::NO_BINADE;
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

}  // namespace

int launch_span_fused(int W, unsigned grid_x, unsigned grid_y, hipStream_t st, const FusedState& fs, uint32_t k, float threshold, int max_it,
                      const uint32_t* d_counts, const float* d_bg, float saturation, float* tables, seqsum::BlockRecord* rec,
                      unsigned long long* look, uint32_t epoch, const uint32_t* bg_range, uint32_t skew, uint32_t lookback) {
#define PENGK_FUSED_CASE(WW)                                                                                                       \
  case WW:                                                                                                                          \
    if constexpr (LookGeo<WW>::SUPPORTED)                                                                                           \
      hipLaunchKernelGGL((em_span_fused_kernel<WW>), dim3(grid_x, grid_y), dim3(FusedGeo::THREADS), 0, st, fs, k, threshold, max_it, \
                         d_counts, d_bg, saturation, tables, rec, look, epoch, bg_range, skew, lookback);                            \
    return PENGK_OK;
  switch (W) {
    PENGK_FUSED_CASE(10)
    PENGK_FUSED_CASE(12)
    default:
      return PENGK_ERR_UNSUPPORTED;
  }
#undef PENGK_FUSED_CASE
}
}  // namespace pengk
