// seqsum.h -- a sequential float32 sum, evaluated in parallel, bit for bit.
//
// The reference adds the 4^(W-1) weights of a PWM cell one after the other in float32 (src/peng.cpp:121-127), and the
// last bits of those sums decide what the program prints (em.hip).  s <- fl(s + t_i) is a chain of dependent roundings,
// but for terms t_i >= 0 it has structure a wave can use:
//
//   * While s stays inside one binade [2^e, 2^(e+1)) its rounding step u = 2^(e-23) is fixed, and
//     fl(s + t) - s depends on t and on the PARITY of s / u only (round-to-nearest-even breaks ties by parity; every
//     other case is decided by t / u alone).  So a stretch of terms acts on every s of the binade as one of two
//     increments, D[parity of s / u], as long as the sum does not leave the binade.
//   * The increments are measured with the float adder itself: run the stretch from the two lowest values of the
//     binade, B0 = 2^e (even) and B1 = 2^e + u (odd).  X_p = B_p + t_0 + t_1 + ... (float additions, in order);
//     D[p] = X_p - B_p, exact.  The true s is >= the base of its parity, so if the true chain stays in the binade the
//     base chain does too, and takes the same increments.
//   * Stretches compose: (A then B)_p = A_p + (B_q - B_q's base), q = parity of A_p = the last mantissa bit of A_p.
//     The composition is associative, so 64 lanes evaluate 64 consecutive stretches and combine them in six steps.
//   * Leaving the binade cannot be missed: all terms are >= 0, so every quantity above is monotone, and a chain that
//     reaches 2^(e+1) anywhere ends >= 2^(e+1) ("flagged").  The wave then finds the first flagged lane with a prefix
//     composition, lets that lane's 64 terms be added the reference's way from the exact value in front of them, and
//     re-evaluates the lanes behind it in the new binade.  A sum crosses at most ~280 binades, in practice a few dozen.
//   * Zero, denormals and the lowest normal binade share u = 2^-149 and are added without rounding: one "binade"
//     [0, 2^-125) with bases 0 and 2^-149.  +inf (overflow) absorbs everything after it.
//
// A term with its sign bit set, an infinity or a NaN breaks the monotonicity; callers route such chains to a plain
// serial loop (em.hip keeps its serial fold kernel for that; fold_chain<.., true> checks and falls back itself).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pengk {
namespace seqsum {

constexpr uint32_t SEG = 64;             // terms per lane and block
constexpr uint32_t BLOCK = 64 * SEG;     // terms per wave and block
constexpr uint32_t SEG_STRIDE = SEG + 4; // floats between the LDS rows of two lanes (272 B: 16-byte reads of all lanes
                                         // fall on different banks)
constexpr uint32_t LDS_FLOATS = 64 * SEG_STRIDE;
constexpr uint32_t INF_BITS = 0x7F800000u;

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t bits(float v) { return __float_as_uint(v); }

struct Bases {
  float b0, b1;     // lowest even / odd value of the binade of s
  uint32_t limit;   // bit pattern of the first value above the binade
};
__device__ __forceinline__ Bases bases_of(float s) {  // s >= +0, finite
  const uint32_t e = bits(s) >> 23;
  Bases r;
  if (e <= 1u) {
    r.b0 = 0.0f;
    r.b1 = __uint_as_float(1u);
    r.limit = 2u << 23;
  } else {
    r.b0 = __uint_as_float(e << 23);
    r.b1 = __uint_as_float((e << 23) | 1u);
    r.limit = (e + 1u) << 23;
  }
  return r;
}

// (a then b): a_p <- a_p + (b_q - base_q), q = parity of a_p
__device__ __forceinline__ void compose(float& a0, float& a1, float b0, float b1, const Bases& B) {
  const float d0 = b0 - B.b0, d1 = b1 - B.b1;
  a0 = a0 + ((bits(a0) & 1u) ? d1 : d0);
  a1 = a1 + ((bits(a1) & 1u) ? d1 : d0);
}

// the 64 terms of an LDS row, added in order to two running values
__device__ __forceinline__ void run2(const float* row, float& x0, float& x1) {
  const f4* q = reinterpret_cast<const f4*>(row);
#pragma unroll
  for (uint32_t j = 0; j < SEG / 4u; ++j) {
    const f4 v = q[j];
    x0 += v.x; x1 += v.x;
    x0 += v.y; x1 += v.y;
    x0 += v.z; x1 += v.z;
    x0 += v.w; x1 += v.w;
  }
}
__device__ __forceinline__ float run1(const float* row, float x) {
  const f4* q = reinterpret_cast<const f4*>(row);
#pragma unroll
  for (uint32_t j = 0; j < SEG / 4u; ++j) {
    const f4 v = q[j];
    x += v.x;
    x += v.y;
    x += v.z;
    x += v.w;
  }
  return x;
}

// One block: lds holds 64 rows of 64 terms (row l = terms 64 l .. 64 l + 63 of the block); s = the sum in front of
// the block.  Returns the sum behind it.  Wave-uniform.
__device__ __forceinline__ float fold_block(const float* lds, uint32_t lane, float s) {
  const float* row = lds + lane * SEG_STRIDE;
  uint32_t first = 0;  // rows < first are already part of s
  for (;;) {
    if (bits(s) >= INF_BITS) return s;  // +inf + t = +inf
    const Bases B = bases_of(s);
    float x0 = B.b0, x1 = B.b1;
    run2(row, x0, x1);
    if (lane < first) {
      x0 = B.b0;
      x1 = B.b1;
    }
    // ordered reduction: after step d lane l (a multiple of 2d) holds rows l .. l + 2d - 1
    float t0 = x0, t1 = x1;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const float b0 = __shfl_down(t0, d, 64), b1 = __shfl_down(t1, d, 64);
      compose(t0, t1, b0, b1, B);
    }
    const float T0 = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)bits(t0)));
    const float T1 = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)bits(t1)));
    const uint32_t par = bits(s) & 1u;
    const float out = s + (par ? T1 - B.b1 : T0 - B.b0);
    if (bits(out) < B.limit) return out;  // the whole block stayed in the binade of s
    // some row leaves the binade: inclusive prefix composition, rows first .. l
    float i0 = x0, i1 = x1;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      float a0 = __shfl_up(i0, d, 64), a1 = __shfl_up(i1, d, 64);
      compose(a0, a1, i0, i1, B);
      if (lane >= (uint32_t)d) {
        i0 = a0;
        i1 = a1;
      }
    }
    const float end = s + (par ? i1 - B.b1 : i0 - B.b0);  // the sum behind row l, exact if nothing crossed up to there
    const unsigned long long flagged = __builtin_amdgcn_ballot_w64(bits(end) >= B.limit);
    const int L = flagged ? __builtin_ctzll(flagged) : 63;  // (flagged != 0: the last row carries the block total)
    float v = s;
    if (L > 0) v = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)bits(end), L - 1));
    s = run1(lds + (uint32_t)L * SEG_STRIDE, v);  // the reference's own additions through row L, all lanes alike
    first = (uint32_t)L + 1u;
    if (first == 64u) return s;
  }
}

// The term source of a chain: load(b, lane, R) fetches block b into 64 registers, deposit(lane, R, lds) spreads them
// over the LDS rows.  fold_chain overlaps the fetch of block b + 1 with the evaluation of block b.
// CHECK: look at every term; a chain with a negative / non-finite term is summed by the plain loop `serial`.
template <class Source, bool CHECK>
__device__ __forceinline__ float fold_chain(const Source& src, uint32_t n_blocks, float* lds, uint32_t lane) {
  float R[64];
  src.load(0u, lane, R);
  float s = 0.0f;
#pragma unroll 1
  for (uint32_t b = 0; b < n_blocks; ++b) {
    if (CHECK) {
      uint32_t m = 0;
#pragma unroll
      for (int k = 0; k < 64; ++k) m = max(m, bits(R[k]));
      if (__builtin_amdgcn_ballot_w64(m > 0x7F7FFFFFu)) return src.serial();  // wave-uniform
    }
    src.deposit(lane, R, lds);
    if (b + 1u < n_blocks) src.load(b + 1u, lane, R);
    s = fold_block(lds, lane, s);
    __builtin_amdgcn_wave_barrier();
  }
  return s;
}

}  // namespace seqsum
}  // namespace pengk
