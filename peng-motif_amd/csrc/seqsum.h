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
// One wave evaluates; a second wave of the workgroup keeps it fed (fold_chain).
// tests/test_seqsum_model.py restates the arithmetic in numpy float32 (CPU); tests/test_gpu_seqsum.py checks this code.
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
typedef __attribute__((address_space(3))) float lds_float;  // (LDS as LDS: global_load_lds takes no generic pointer)
typedef __attribute__((address_space(3))) f4 lds_f4;

#ifdef PENGK_SEQSUM_STATS  // developer build: where a chain's time goes (tools/seqsum_stats.py)
__device__ unsigned long long g_stats[12];
struct Stats {  // per wave, added to g_stats once at the end of the chain
  unsigned long long v[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  __device__ __forceinline__ void flush(uint32_t lane) {
    if (lane == 0)
      for (int i = 0; i < 12; ++i) atomicAdd(&g_stats[i], v[i]);
  }
};
#define PENGK_STAT_ADD(i, x) do { st.v[i] += (unsigned long long)(x); } while (0)
#define PENGK_CLOCK() __builtin_amdgcn_s_memtime()
#else
struct Stats {
  __device__ __forceinline__ void flush(uint32_t) {}
};
#define PENGK_STAT_ADD(i, x) do { } while (0)
#define PENGK_CLOCK() 0ull
#endif

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

// An LDS row's 64 terms into registers: all sixteen 16-byte reads are issued before the first value is used (left to
// itself the compiler keeps two reads in flight and the chain of additions waits for LDS eight times per row).
struct Row {
  f4 q[SEG / 4u];
  __device__ __forceinline__ void read(const float* row) {
    const f4* src = reinterpret_cast<const f4*>(row);
#pragma unroll
    for (uint32_t j = 0; j < SEG / 4u; ++j) q[j] = src[j];
    static_assert(SEG == 64u, "sixteen quads");
    asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]), "+v"(q[8]),
                 "+v"(q[9]), "+v"(q[10]), "+v"(q[11]), "+v"(q[12]), "+v"(q[13]), "+v"(q[14]), "+v"(q[15]));
  }
  // a row of a block that Source::stage put into LDS: 256-byte rows, slot c of row l at c ^ (l & 15)
  __device__ __forceinline__ void read_staged(const lds_float* buf, uint32_t l) {
    const lds_f4* src = (const lds_f4*)buf + 16u * l;
#pragma unroll
    for (uint32_t j = 0; j < SEG / 4u; ++j) q[j] = src[j ^ (l & 15u)];
    asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]), "+v"(q[8]),
                 "+v"(q[9]), "+v"(q[10]), "+v"(q[11]), "+v"(q[12]), "+v"(q[13]), "+v"(q[14]), "+v"(q[15]));
  }
  // Rounded float32 additions, x <- fl(x + t), written as the instructions themselves, sixteen terms per statement:
  // left to the optimiser the two chains of run2 become v_pk_add_f32 on (t, t) pairs -- 128 register copies per block to
  // build the pairs, half-rate packed additions, twice the registers (206 instead of ~100: two waves per SIMD instead of
  // four); and one statement per addition makes the hazard recogniser put an s_nop behind every one of them.
#define PENGK_ADD2(T) "v_add_f32 %0, " T ", %0\n\tv_add_f32 %1, " T ", %1\n\t"
#define PENGK_ADD1(T) "v_add_f32 %0, " T ", %0\n\t"
  static __device__ __forceinline__ void add16x2(float& x0, float& x1, const f4& a, const f4& b, const f4& c, const f4& d) {
    asm(PENGK_ADD2("%2") PENGK_ADD2("%3") PENGK_ADD2("%4") PENGK_ADD2("%5") PENGK_ADD2("%6") PENGK_ADD2("%7") PENGK_ADD2("%8")
        PENGK_ADD2("%9") PENGK_ADD2("%10") PENGK_ADD2("%11") PENGK_ADD2("%12") PENGK_ADD2("%13") PENGK_ADD2("%14")
        PENGK_ADD2("%15") PENGK_ADD2("%16") PENGK_ADD2("%17")
        : "+v"(x0), "+v"(x1)
        : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w), "v"(c.x), "v"(c.y), "v"(c.z), "v"(c.w),
          "v"(d.x), "v"(d.y), "v"(d.z), "v"(d.w));
  }
  static __device__ __forceinline__ void add16(float& x, const f4& a, const f4& b, const f4& c, const f4& d) {
    asm(PENGK_ADD1("%1") PENGK_ADD1("%2") PENGK_ADD1("%3") PENGK_ADD1("%4") PENGK_ADD1("%5") PENGK_ADD1("%6") PENGK_ADD1("%7")
        PENGK_ADD1("%8") PENGK_ADD1("%9") PENGK_ADD1("%10") PENGK_ADD1("%11") PENGK_ADD1("%12") PENGK_ADD1("%13")
        PENGK_ADD1("%14") PENGK_ADD1("%15") PENGK_ADD1("%16")
        : "+v"(x)
        : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w), "v"(c.x), "v"(c.y), "v"(c.z), "v"(c.w),
          "v"(d.x), "v"(d.y), "v"(d.z), "v"(d.w));
  }
#undef PENGK_ADD2
#undef PENGK_ADD1
  // the terms added in order to two running values / to one
  __device__ __forceinline__ void run2(float& x0, float& x1) const {
#pragma unroll
    for (uint32_t j = 0; j < SEG / 4u; j += 4u) add16x2(x0, x1, q[j], q[j + 1u], q[j + 2u], q[j + 3u]);
  }
  __device__ __forceinline__ float run1(float x) const {
#pragma unroll
    for (uint32_t j = 0; j < SEG / 4u; j += 4u) add16(x, q[j], q[j + 1u], q[j + 2u], q[j + 3u]);
    return x;
  }
};

// v of the lane d places down in this lane's row of 16 (DPP row_shr), `self` where the row has no such lane
template <int D>
__device__ __forceinline__ float row_shr(float self, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(self), __float_as_int(v), 0x110 + D, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_value(float v, int l) {  // wave-uniform
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// i0, i1 = what this lane's stretch makes of the two bases of B (the bases themselves for a lane that has nothing to add);
// s = the sum in front of lane 0's stretch, in the binade of B.  Returns the sum behind this lane's stretch -- exact if
// nothing left the binade up to there, >= 2^(e+1) otherwise.  An inclusive prefix composition: four DPP steps inside the
// rows of 16 lanes, the three row boundaries with wave-uniform values; a plain prefix sum when no stretch met a tie.
__device__ __forceinline__ float ends_behind(float i0, float i1, const Bases& B, float s, uint32_t lane) {
  const uint32_t par = bits(s) & 1u;
  const uint32_t rw = lane >> 4;
  float end;
  const float d0 = i0 - B.b0, d1 = i1 - B.b1;
  if (__builtin_amdgcn_ballot_w64(bits(d0) != bits(d1)) == 0ull) {
    // No row met a tie: both parities take the same increment, and the composition is a prefix SUM of increments --
    // multiples of u below 2^24 u, so the float additions are exact (or the result is >= 2^(e+1) and gets flagged).
    float d = d0;
    d += row_shr<1>(0.0f, d);
    d += row_shr<2>(0.0f, d);
    d += row_shr<4>(0.0f, d);
    d += row_shr<8>(0.0f, d);
    // across the rows of 16: lane 15 of a row into every lane of the next row (rows 1 and 3), then lane 31 -- by now
    // the total of rows 0 and 1 -- into rows 2 and 3: the wave64 scan idiom of gfx9's DPP broadcasts
    d += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x142 /* row_bcast:15 */, 0xa, 0xf, false));
    d += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x143 /* row_bcast:31 */, 0xc, 0xf, false));
    end = s + d;
  } else {
    // inclusive prefix inside each row of 16 lanes: (rows l-d .. of the earlier lanes) then (this lane's)
#define PENGK_ROW_STEP(D)                                                     \
    {                                                                        \
      float a0 = row_shr<D>(B.b0, i0), a1 = row_shr<D>(B.b1, i1);            \
      compose(a0, a1, i0, i1, B);                                            \
      i0 = a0;                                                               \
      i1 = a1;                                                               \
    }
    PENGK_ROW_STEP(1) PENGK_ROW_STEP(2) PENGK_ROW_STEP(4) PENGK_ROW_STEP(8)
#undef PENGK_ROW_STEP
    // the rows of 16: totals at lanes 15, 31, 47; what lies in front of rows 1, 2, 3 (wave-uniform)
    float p0 = lane_value(i0, 15), p1 = lane_value(i1, 15);  // in front of row 1
    float q0 = p0, q1 = p1;
    compose(q0, q1, lane_value(i0, 31), lane_value(i1, 31), B);  // in front of row 2
    float r0 = q0, r1 = q1;
    compose(r0, r1, lane_value(i0, 47), lane_value(i1, 47), B);  // in front of row 3
    float f0 = rw == 1u ? p0 : rw == 2u ? q0 : rw == 3u ? r0 : B.b0;
    float f1 = rw == 1u ? p1 : rw == 2u ? q1 : rw == 3u ? r1 : B.b1;
    compose(f0, f1, i0, i1, B);  // rows first .. this lane's
    end = s + (par ? f1 - B.b1 : f0 - B.b0);
  }
  return end;
}

// One block: `mine` holds this lane's row of 64 terms (row l = terms 64 l .. 64 l + 63 of the block); s = the sum in
// front of the block.  Returns the sum behind it.  Wave-uniform.
//
// Every lane evaluates its row from the two bases; an inclusive prefix composition (four DPP steps inside the rows of
// 16 lanes, the three row boundaries with wave-uniform values) gives every lane the sum behind its row, valid if nothing
// left the binade up to there.  No lane flagged: lane 63 holds the result.  Otherwise the first flagged row is added
// the reference's way and the rows behind it are evaluated again in the new binade.
// The block lives in registers for the whole call -- LDS is not touched, so the fetching wave may overwrite the
// buffer with the next block while this one is evaluated (fold_chain).
__device__ __forceinline__ float fold_block(const Row& mine, uint32_t lane, float s, Stats& st) {
  uint32_t first = 0;  // rows < first are already part of s
  for (;;) {
    PENGK_STAT_ADD(1, 1);
    const unsigned long long k2 = PENGK_CLOCK();
    if (bits(s) >= INF_BITS) return s;  // +inf + t = +inf
    const Bases B = bases_of(s);
    float i0 = B.b0, i1 = B.b1;
    mine.run2(i0, i1);
#ifdef PENGK_SEQSUM_STATS
    asm volatile("" : "+v"(i0), "+v"(i1));
#endif
    const unsigned long long k3 = PENGK_CLOCK();
    PENGK_STAT_ADD(5, k3 - k2);
    if (lane < first) {
      i0 = B.b0;
      i1 = B.b1;
    }
    const float end = ends_behind(i0, i1, B, s, lane);  // the sum behind this lane's row, exact if nothing crossed up to there
    const unsigned long long flagged = __builtin_amdgcn_ballot_w64(bits(end) >= B.limit);
    const unsigned long long k4 = PENGK_CLOCK();
    PENGK_STAT_ADD(6, k4 - k3);
    if (!flagged) return lane_value(end, 63);  // the whole block stayed in the binade of s
    const int L = __builtin_ctzll(flagged);
    float v = s;
    if (L > 0) v = lane_value(end, L - 1);
    // the reference's own additions through row L from the exact value in front of it: every lane adds ITS row to v,
    // lane L's result is the one that counts (no second copy of a row, no LDS)
    s = lane_value(mine.run1(v), L);
    first = (uint32_t)L + 1u;
#ifdef PENGK_SEQSUM_STATS
    asm volatile("" : "+v"(s));
#endif
    PENGK_STAT_ADD(7, PENGK_CLOCK() - k4);
    if (first == 64u) return s;
  }
}

// ---- blocks ahead of their chain ------------------------------------------------------------------------------------
// What a block costs -- 128 additions per lane, the prefix composition -- depends on the BINADE of the sum in front of it
// only, not on the sum: X_p = (B_p then the block's 4096 terms) for the two bases of binade e is everything the chain
// needs from a block it passes without leaving e.  So the blocks of a chain can be evaluated all at once, each under the
// binade a cheap estimate of the sum in front of it predicts (em.hip: block sums, their prefix), and the chain itself
// becomes one addition per block: s <- s + (X_par - B_par), valid if s lies in e and the result stays below 2^(e+1) --
// the same test fold_block applies to its lane 63.  A block whose binade was not predicted (the estimate too close to a
// power of two, the sum about to cross one) or whose test fails is evaluated by fold_block as before.  Exactness never
// rests on the estimate: it only decides which blocks take the short way.
constexpr uint32_t NO_BINADE = 0xFFFFFFFFu;
// record of a chain's block 0 when the chain's first blocks were folded ahead of it (the start is known exactly: zero):
// d0 = the sum behind them, pad = how many (1..64; their own records hold nothing)
constexpr uint32_t SUM_BEHIND = 0xFFFFFFFEu;
// exponent field of the binade of s as bases_of sees it: zero, denormals and the lowest normal binade are one
__device__ __forceinline__ uint32_t binade_of(float s) {
  const uint32_t e = bits(s) >> 23;
  return e <= 1u ? 1u : e;
}
__device__ __forceinline__ Bases bases_of_binade(uint32_t e) {  // e = binade_of(s) of a finite s
  Bases r;
  r.b0 = __uint_as_float(e <= 1u ? 0u : e << 23);
  r.b1 = __uint_as_float(e <= 1u ? 1u : (e << 23) | 1u);
  r.limit = (e + 1u) << 23;
  return r;
}

// The block in `mine` (row l = this lane's 64 terms) run from both bases of B: the increments D[p] = X_p - B_p, wave-
// uniform; false if a base chain reaches the end of the binade (then nothing is known about the block).
__device__ __forceinline__ bool block_increments(const Row& mine, uint32_t lane, const Bases& B, float& D0, float& D1) {
  float i0 = B.b0, i1 = B.b1;
  mine.run2(i0, i1);
  const uint32_t rw = lane >> 4;
  const float d0 = i0 - B.b0, d1 = i1 - B.b1;
  float x0, x1;
  if (__builtin_amdgcn_ballot_w64(bits(d0) != bits(d1)) == 0ull) {  // (as in fold_block: a prefix SUM of increments)
    float d = d0;
    d += row_shr<1>(0.0f, d);
    d += row_shr<2>(0.0f, d);
    d += row_shr<4>(0.0f, d);
    d += row_shr<8>(0.0f, d);
    d += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x142 /* row_bcast:15 */, 0xa, 0xf, false));
    d += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x143 /* row_bcast:31 */, 0xc, 0xf, false));
    x0 = B.b0 + d;
    x1 = B.b1 + d;
  } else {
#define PENGK_ROW_STEP(D)                                                     \
    {                                                                          \
      float a0 = row_shr<D>(B.b0, i0), a1 = row_shr<D>(B.b1, i1);              \
      compose(a0, a1, i0, i1, B);                                              \
      i0 = a0;                                                                 \
      i1 = a1;                                                                 \
    }
    PENGK_ROW_STEP(1) PENGK_ROW_STEP(2) PENGK_ROW_STEP(4) PENGK_ROW_STEP(8)
#undef PENGK_ROW_STEP
    float p0 = lane_value(i0, 15), p1 = lane_value(i1, 15);
    float q0 = p0, q1 = p1;
    compose(q0, q1, lane_value(i0, 31), lane_value(i1, 31), B);
    float r0 = q0, r1 = q1;
    compose(r0, r1, lane_value(i0, 47), lane_value(i1, 47), B);
    float f0 = rw == 1u ? p0 : rw == 2u ? q0 : rw == 3u ? r0 : B.b0;
    float f1 = rw == 1u ? p1 : rw == 2u ? q1 : rw == 3u ? r1 : B.b1;
    compose(f0, f1, i0, i1, B);
    x0 = f0;
    x1 = f1;
  }
  x0 = lane_value(x0, 63);
  x1 = lane_value(x1, 63);
  D0 = x0 - B.b0;
  D1 = x1 - B.b1;
  // (every value above is monotone in the terms: a base chain that reached 2^(e+1) anywhere ends there or above)
  return bits(x0) < B.limit && bits(x1) < B.limit;
}

// One evaluated block of a chain: {binade e or NO_BINADE, D[0], D[1]} (16 bytes: one load per lane and 64 blocks)
struct BlockRecord {
  uint32_t e;
  float d0, d1;
  uint32_t pad;
};

// The chain of a cell over evaluated blocks (one wave).  rec[b] as block_increments left it; n_blocks is a multiple of 64;
// first_records = rec[lane] (the caller loads it beside its own flags).
// The blocks without a binade come from the table, through LDS: Source::stage(b, lane, buf) asks for block b with loads
// that write LDS directly (16 KiB, rows of 256 bytes, Row::read_staged).  No register waits for them, so TWO blocks are
// on their way at any time and the wait for the earlier one is `s_waitcnt vmcnt(loads of the later one)`: walking the
// evaluated blocks between two fetched ones, and fold_block on one, take less time than memory does (~3 us for a
// block another kernel wrote), and with loads into registers -- where the compiler places the waits, vmcnt(0) whenever
// it cannot count -- the chain stood waiting at every fetched block.  lds = two buffers of BLOCK floats.
constexpr uint32_t WALK_LDS_FLOATS = 2u * BLOCK;
// -DPENGK_TEST_WALK_RACE re-creates round 3's defect (the counted wait also for a re-requested block): a developer build
// that exists to show that test_em_chain_walk_restaged_block_is_pinned fails on it; never part of the product build.
#ifdef PENGK_TEST_WALK_RACE
#define PENGK_WALK_RESTAGED(x) false
#else
#define PENGK_WALK_RESTAGED(x) (x)
#endif
template <uint32_t N>
__device__ __forceinline__ void wait_loads_but() {  // until at most N of this wave's loads are outstanding
  static_assert(N == 0u || N == 16u, "the counts walk_chain needs");
  if (N == 0u)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
}
// What a chain met on its way (wave-uniform; the kernel adds them up per pengk_em call, pengk_get_info "em_*"):
//   fetched      blocks taken the long way (fold_block on the block's terms): the ones without a binade + the mispredicted
//   mispredicted blocks that HAD a binade which did not hold when the chain got there: fetched on demand
//   restaged     ... through the buffer of a block already asked for, which is then asked for again
//   restaged_waits  takes of such a re-requested block while the other buffer's (older) loads were still counted: the
//                   wait that must be vmcnt(0), not the counted one (the round-3 defect: tests/test_gpu_parity.py pins it)
struct WalkCounts {
  uint32_t fetched = 0, mispredicted = 0, restaged = 0, restaged_waits = 0;
};
template <class Source>
__device__ __forceinline__ float walk_chain(const Source& src, const BlockRecord* __restrict__ rec, const uint4& first_records,
                                            uint32_t n_blocks, lds_float* lds, uint32_t lane, WalkCounts& wc) {
  Stats st;
  const unsigned long long w0 = PENGK_CLOCK();
  float s = 0.0f;
  lds_float* bufa = lds;
  lds_float* bufb = lds + BLOCK;
  uint32_t ha = NO_BINADE, hb = NO_BINADE;  // the blocks in bufa / bufb (or on their way)
  bool a_first = true;                      // which of the two comes first in the chain
  bool restaged = false;                    // that one was asked for again after the other: the counted wait does not cover it
  uint4 r_next = first_records;  // (this lane's record of the first chunk, loaded by the caller together with what else it needs)
#pragma unroll 1
  for (uint32_t base = 0; base < n_blocks; base += 64u) {
    // The next chunk's records are asked for while this one is walked (W = 12: sixteen chunks per chain, each a dependent
    // round trip in front of its walk).  The load is older than every block this chunk stages, so the counted waits
    // below still cover what they name.
    const uint4 r = r_next;
    if (base + 64u < n_blocks) r_next = reinterpret_cast<const uint4*>(rec)[base + 64u + lane];
    unsigned long long open = __builtin_amdgcn_ballot_w64(r.x == NO_BINADE);
    auto open_after = [&](uint32_t b) {  // the next block of this chunk without a binade behind block b, or NO_BINADE
      const uint32_t j = b - base;
      const unsigned long long rest = j < 63u ? open & ~((2ull << j) - 1ull) : 0ull;
      return rest ? base + (uint32_t)__builtin_ctzll(rest) : NO_BINADE;
    };
    uint32_t j = 0;
    if (base == 0u && (uint32_t)__builtin_amdgcn_readfirstlane((int)r.x) == SUM_BEHIND) {
      // the first blocks were folded from zero by the kernel that evaluated the blocks (em_span_eval_kernel's extra
      // workgroups): record 0 carries the sum behind them and their number (1..64); their own records hold nothing
      s = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)r.y));
      j = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.w);
      j = j < 1u ? 1u : j > 64u ? 64u : j;
      open = j < 64u ? open & ~((1ull << j) - 1ull) : 0ull;
    }
    if (open) {  // (nothing is in flight across chunks)
      ha = base + (uint32_t)__builtin_ctzll(open);
      src.stage(ha, lane, bufa);
      hb = open_after(ha);
      if (hb != NO_BINADE) src.stage(hb, lane, bufb);
      a_first = true;
      restaged = false;
    }
#pragma unroll 1
    while (j < 64u) {
      // the evaluated blocks from j on that share the binade of s: their increments compose like the rows of a block
      // (ends_behind, lanes = blocks); the first one whose sum would leave the binade -- if any -- ends the run
      const uint32_t e = binade_of(s);
      const unsigned long long same = __builtin_amdgcn_ballot_w64(r.x == e) >> j;
      const uint32_t len = same == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~same);
      if (len != 0u && bits(s) < INF_BITS) {
        const Bases B = bases_of_binade(e);
        const bool in = lane >= j && lane < j + len;
        const float end = ends_behind(in ? B.b0 + __uint_as_float(r.y) : B.b0, in ? B.b1 + __uint_as_float(r.z) : B.b1, B, s, lane);
        const unsigned long long flagged = __builtin_amdgcn_ballot_w64(in && bits(end) >= B.limit);
        if (!flagged) {
          s = lane_value(end, (int)(j + len - 1u));
          j += len;
          continue;
        }
        const uint32_t L = (uint32_t)__builtin_ctzll(flagged);
        if (L > j) s = lane_value(end, (int)(L - 1u));
        j = L;  // block L is where the sum leaves the binade: evaluated the long way
      }
      const uint32_t b = base + j;
      const unsigned long long c0 = PENGK_CLOCK();
      Row mine;
      // buf1 / h1 = the block asked for first.  Block b is that one -- then its buffer is free once the rows are in
      // registers, the block behind the later one is asked for, and the later one becomes the first -- or b is a block
      // whose binade did not hold: fetched now, through buf1, whose own block is asked for again (and is then the
      // YOUNGER request of the two: only vmcnt(0) covers it).
#define PENGK_TAKE(buf1, h1, h2)                                                          \
      {                                                                                   \
        const bool ahead = h1 == b;                                                       \
        if (!ahead) {                                                                     \
          src.stage(b, lane, buf1);                                                       \
          wait_loads_but<0u>();                                                           \
          PENGK_STAT_ADD(9, 1);                                                           \
          ++wc.mispredicted;                                                              \
        } else if (h2 != NO_BINADE && !PENGK_WALK_RESTAGED(restaged) && Source::STAGE_LOADS == 16u) { \
          wait_loads_but<16u>();                                                          \
        } else {                                                                          \
          wc.restaged_waits += (h2 != NO_BINADE && restaged) ? 1u : 0u;                   \
          wait_loads_but<0u>();                                                           \
        }                                                                                 \
        mine.read_staged(buf1, lane);                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                \
        if (ahead) {                                                                      \
          h1 = open_after(h2 != NO_BINADE ? h2 : b);                                      \
          a_first = !a_first;                                                             \
          restaged = false;                                                               \
        } else {                                                                          \
          restaged = h1 != NO_BINADE; /* its loads are now YOUNGER than the other buffer's */ \
          wc.restaged += restaged ? 1u : 0u;                                              \
        }                                                                                 \
        if (h1 != NO_BINADE) src.stage(h1, lane, buf1);                                   \
      }
      if (a_first) PENGK_TAKE(bufa, ha, hb) else PENGK_TAKE(bufb, hb, ha)
#undef PENGK_TAKE
      const unsigned long long c1 = PENGK_CLOCK();
      PENGK_STAT_ADD(2, c1 - c0);
      s = fold_block(mine, lane, s, st);
#ifdef PENGK_SEQSUM_STATS
      asm volatile("" : "+v"(s));
#endif
      PENGK_STAT_ADD(3, PENGK_CLOCK() - c1);
      PENGK_STAT_ADD(0, 1);
      ++wc.fetched;
      ++j;
    }
  }
#ifdef PENGK_SEQSUM_STATS
  asm volatile("" : "+v"(s));
#endif
  PENGK_STAT_ADD(8, PENGK_CLOCK() - w0);
  PENGK_STAT_ADD(10, 1);
  st.flush(lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (nothing may still be writing LDS when the workgroup's LDS is given back)
  return s;
}

// The term source of a chain, for NF fetching waves: load<NF>(b, part, lane, R) fetches share `part` of block b into
// 64 / NF registers, deposit<NF>(part, lane, R, lds) spreads them over the LDS rows.
//
// 1 + FETCH_WAVES waves per chain (a workgroup of CHAIN_THREADS threads, every thread calls fold_chain): moving a
// block through a wave -- sixteen 16-byte loads, sixteen LDS writes, the wait for them -- takes as long as evaluating
// it, so the waves behind wave 0 only fetch (whole blocks, in turn), while wave 0 only evaluates.
// ONE row buffer in LDS (17 KiB: eight chains per CU, two evaluating waves per SIMD that fill each other's waits):
//   barrier B(i)   block i is in LDS            wave 0 reads its rows into registers (the whole block: 64 x 64 terms)
//   barrier A(i)   block i is in registers      the fetchers deposit block i + 1 (its loads were issued during block
//                                               i - 1) while wave 0 evaluates block i from registers
// The result is returned in wave 0 (the others return 0).
// CHECK: the fetching waves look at every term; a chain with a negative / non-finite term is summed by the plain loop
// `serial`.  The flag is written in front of B(i) and read by every wave between B(i) and A(i); the next write comes
// behind A(i): all waves leave at the same i.
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's GLOBAL loads (s_waitcnt
// vmcnt(0) in front of s_barrier): a fetch wave that has just issued the loads of a later block would stand at the
// barrier until they have landed, and everybody with it -- the memory latency the prefetch is there to hide.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifndef PENGK_FETCH_WAVES
#define PENGK_FETCH_WAVES 1
#endif
// FETCH_WAVES fetching waves take the blocks in turn (wave q moves the blocks i = q mod FETCH_WAVES, whole): the loads
// of a block are issued FETCH_WAVES block times before it is needed.  Measured with the LDS-only barrier, i.e. with
// loads that really stay in flight across steps (16 PWMs x 10 iterations, W = 10): 1 -> 0.94 ms, 2 -> 0.98, 3 -> 0.99,
// 4 -> 1.32 -- a step is NOT waiting for its loads; it takes what the evaluating wave's dependent additions take
// (profiles/r03_em_experiments.log).
constexpr uint32_t FETCH_WAVES = PENGK_FETCH_WAVES;
constexpr uint32_t CHAIN_THREADS = 64u * (1u + FETCH_WAVES);
constexpr uint32_t CHAIN_LDS_FLOATS = LDS_FLOATS + 4u;  // the row buffer + the flag word
template <class Source, bool CHECK>
__device__ __forceinline__ float fold_chain(const Source& src, uint32_t n_blocks, float* lds, uint32_t thread) {
  const uint32_t lane = thread & 63u;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(thread >> 6));  // wave-uniform
  const bool fetcher = wave != 0u;
  const uint32_t q = wave - 1u;  // a fetch wave's turn
  volatile uint32_t* bad = reinterpret_cast<volatile uint32_t*>(lds + LDS_FLOATS);
  if (CHECK) {
    if (thread == 0) *bad = 0u;
    __syncthreads();
  }
  Stats st;
  float s = 0.0f;
  bool fallback = false;
  if (fetcher) {
    float R[64];
    if (q < n_blocks) src.template load<1>(q, 0u, lane, R);
#pragma unroll 1
    for (uint32_t i = 0; i < n_blocks; ++i) {
      if (i % FETCH_WAVES == q) {
        if (CHECK) {
          uint32_t m = 0;
#pragma unroll
          for (int k = 0; k < 64; ++k) m = max(m, bits(R[k]));
          if (__builtin_amdgcn_ballot_w64(m > 0x7F7FFFFFu) && lane == 0) *bad = 1u;
        }
        const unsigned long long c0 = PENGK_CLOCK();
        src.template deposit<1>(0u, lane, R, lds);
        if (i + FETCH_WAVES < n_blocks) src.template load<1>(i + FETCH_WAVES, 0u, lane, R);
        PENGK_STAT_ADD(2, PENGK_CLOCK() - c0);
      }
      lds_barrier();  // B(i)
      if (CHECK && *bad) break;
      lds_barrier();  // A(i)
    }
  } else {
#pragma unroll 1
    for (uint32_t i = 0; i < n_blocks; ++i) {
      lds_barrier();  // B(i)
      if (CHECK && *bad) {
        fallback = true;
        break;
      }
      const unsigned long long k0 = PENGK_CLOCK();
      Row mine;
      mine.read(lds + lane * SEG_STRIDE);
      PENGK_STAT_ADD(4, PENGK_CLOCK() - k0);
      lds_barrier();  // A(i)
      const unsigned long long c0 = PENGK_CLOCK();
      s = fold_block(mine, lane, s, st);
      PENGK_STAT_ADD(0, 1);
      PENGK_STAT_ADD(3, PENGK_CLOCK() - c0);
    }
    if (CHECK && fallback) s = src.serial();
  }
  st.flush(lane);
  return fetcher ? 0.0f : s;
}

}  // namespace seqsum
}  // namespace pengk
