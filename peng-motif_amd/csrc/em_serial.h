// em_serial.h -- what the kernels of the serial (bit-exact) EM's blocks-ahead scheme share: the span geometry, the binade
// estimate, the lean division, the finalize step at the head of an iteration, the span's LDS layout.  Included by em.hip
// (the three-launch scheme: the product) and em_fused.hip (the opt-in two-launch variant, em_serial_scan = 3).
#pragma once
#include "em_common.h"

namespace pengk {
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

// em_fused.hip: one launch of em_span_fused_kernel<W> (W = 10, 12; other W: PENGK_ERR_UNSUPPORTED) on `st`
int launch_span_fused(int W, unsigned grid_x, unsigned grid_y, hipStream_t st, const FusedState& fs, uint32_t k, float threshold, int max_it,
                      const uint32_t* d_counts, const float* d_bg, float saturation, float* tables, seqsum::BlockRecord* rec,
                      unsigned long long* look, uint32_t epoch, const uint32_t* bg_range, uint32_t skew, uint32_t lookback);

namespace {
// ---- the scan with its blocks evaluated ahead of the chain (seqsum.h, "blocks ahead of their chain"; W >= 10) ----------
// Per iteration and PWM (three launches, nothing between them but the kernel boundaries):
//   em_weights_span_kernel   at its head the PREVIOUS iteration's finalize step -- row normalisation, change, "still
//                            running" (fused_head: every workgroup of a PWM repeats the 4 W divisions from the cell sums
//                            the chains stored; no arrival protocol at a chain's end, no finalize launch) --, then
//                            the weights, span by span, and with them the plain sum of every block of every cell;
//   (em_block_predict_kernel their prefix per cell = an estimate of the sum in front of each block; a block whose estimate
//                            stays clear of a power of two from its first to its last term gets that binade -- for cells of
//                            up to 1024 blocks the next kernel does that itself)
//   em_span_eval_kernel      every block with a binade gets its two increments (block_increments) -- all blocks of all
//                            cells at once, a workgroup per span of the table, instead of one after the other per cell;
//   em_chain_store_kernel    one wave per cell walks the blocks: one addition per evaluated block, fold_block for the
//                            others (the few where the sum crosses a power of two), and stores the cell's sum;
//   (em_serial_finish_kernel once per call: the last iteration's finalize step, every PWM's matrix into the caller's array).
// Option em_serial_scan = 3 (W = 10, 12) runs the first two as ONE kernel, em_span_fused_kernel, below.
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

// Which (PWM, unit) a workgroup takes -- unit = a span, a chain, a group of block-0 cells; U of them per PWM.  PWMs in whole
// groups of eight keep one XCD each (workgroup i runs on XCD i mod 8: a PWM's spans, and behind them its chains, stay with
// one L2; 1000 PWMs: 35.5 against 37.5 ms).  The PWMs of a last, partial group are dealt unit by unit over ALL XCDs
// instead: pinned to one XCD each, a batch of 2 PWMs ran its evaluation on a quarter of the chip (W = 12, 2 PWMs -- what a
// rank of the 8-GPU configs[3] run holds: evaluation kernel 205 us beside a 68 us weights kernel).
__device__ __forceinline__ bool group_map(uint32_t idx, uint32_t n_pwm, uint32_t U, uint32_t& pw, uint32_t& unit) {
  const uint32_t full = n_pwm >> 3, rem = n_pwm & 7u, full_wgs = full * 8u * U;
  if (idx < full_wgs) {
    pw = (idx & 7u) + 8u * ((idx >> 3) / U);
    unit = (idx >> 3) % U;
    return true;
  }
  const uint32_t j = idx - full_wgs;
  if (j >= rem * U) return false;
  pw = 8u * full + j % rem;
  unit = j / rem;
  return true;
}

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

// per-call counters of what the chains met (pengk_get_info "em_*"): fetched, mispredicted, restaged, restaged_waits
// (seqsum::WalkCounts); behind them in the same allocation: the background table's {min, max}
constexpr int EM_COUNTERS = 4;
struct FusedGeo {
  static constexpr uint32_t THREADS = 512, WAVES = 8, CHUNK = 64;
  static constexpr unsigned long long LOOKBACK_TICKS = 50000ull;  // 500 us
};
template <int W>
struct LookGeo {
  using G = BlockGeo<W>;
  static constexpr uint32_t CHUNKS = (G::SPANS + FusedGeo::CHUNK - 1u) / FusedGeo::CHUNK;
  static constexpr bool SUPPORTED = CHUNKS <= 64u;  // two levels: the earlier spans of a chunk, the earlier chunks (W = 14 would take a third)
  static constexpr size_t WORDS_PER_PWM = (size_t)(G::SPANS + CHUNKS) * G::CELLS;  // A[span][cell] | T[chunk][cell]
};
}  // namespace
}  // namespace pengk
