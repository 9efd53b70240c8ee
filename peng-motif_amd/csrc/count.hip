// count.hip -- K1: the 4^W k-mer count over the 2-bit packed stream (gfx950).
//
// Replaces BasePattern::count_patterns / count_patterns_single_strand
// (src/base_pattern.cpp:331-441), the count mirror (:387-392), the background (k+1)-mer count
// (src/shared/BackgroundModel.cpp:60-84) and BackgroundModel::calculateV (:490-530).
//
// Work decomposition: one lane per scan item (<= item_windows windows of one visited run; the
// host packer, pack.cpp, has already resolved the N/skip scan rule).  A lane walks its windows
// left to right with a rolling id / reverse-complement id (2 + 3 ops per base) and keeps the
// canonical ids of the last W-1 COUNTED windows in a 16-slot register ring; the reference's
// "last counted occurrence >= W positions back" rule (:361-366) is then exactly
// "canonical id not in the ring" -- no 4^W last-position table, no second pass.
//
// An item that continues a run (long sequences are split) first replays a 3W-3..-base prologue in
// front of its first window.  If no window of the prologue is suppressed, the ring it leaves is
// provably the true one (see DESIGN.md "non-overlap rule"); otherwise the item is pushed on a
// defer list and redone by count_fixup_kernel, which searches backwards for a certified start.
#include <stdlib.h>

#include <type_traits>

#include "pengk_internal.h"

namespace pengk {

namespace {

constexpr uint32_t INVALID_ID = 0xFFFFFFFFu;  // ids are < 4^14

__device__ __forceinline__ uint32_t funnel(uint32_t hi, uint32_t lo, uint32_t shift) {
  return __builtin_amdgcn_alignbit(hi, lo, shift);  // ((hi:lo) >> shift)[31:0], shift < 32
}

template <int W>
struct Geo {
  static constexpr int P = ((3 * W - 3 + 15) / 16) * 16;  // prologue bases in front of a continuing item
  static constexpr uint32_t MASK = (1u << (2 * W)) - 1u;
  static constexpr int TOP = 2 * (W - 1);
  // main scan: windows ending on bases 0 .. W-2 of a chunk straddle two chunks; a 32-bit view of the pair
  // at bit offset 2(u0 + 17 - W) holds the windows u0 .. u0 + PER - 1 completely
  static constexpr int PER = (32 - 2 * W) / 2 + 1;
  static constexpr int NVIEW = (W - 1 + PER - 1) / PER;
};

// 16 bases (first base in the low bits) -> their reverse complement, first base in the low bits
__device__ __forceinline__ uint32_t revcomp16(uint32_t x) {
  const uint32_t r = __builtin_bitreverse32(x);  // base order reversed, but so are the two bits of every base
  return ~(((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1));
}

// ---------------------------------------------------------------------------------------------
// The scan shared by both K1 variants.  `Emit` receives every window's key: emit.full(key) from all lanes (key = INVALID_ID for a
// suppressed window) or emit.masked(key, active).
// It is called by all lanes of the wave in lock step (active = this lane has a counted window).
// ---------------------------------------------------------------------------------------------
// Fused K1b: while the scan rolls the id, the top three digits ARE the 3-mer ending at the current base.
// Every item owns the bases its windows end on; the first item of a run also owns the run's first W-1
// bases (counted in the prologue).  Bins live in LDS per wave: [0..63] 3-mers (little-endian digits),
// [64..67] first base of a run, [68..83] first 2-mer (x0 | x1 << 2).  Only meaningful for inputs made
// of whole sequences (pengk_packed.all_whole), where runs == sequences.
// Full chunks (every lane owns all 16 window-end bases) count one 4-MER per two bases instead: the 4-mer ending on
// base u holds the 3-mers ending on u-1 and on u, so bins4 is folded into the 3-mer bins at the end of the kernel
// (one LDS add per two bases).
struct BgLds {
  uint32_t bins[96];
  uint32_t bins4[256];
};
__device__ __forceinline__ BgLds& bg_lds() {
  __shared__ BgLds sh;
  return sh;
}
template <int W, bool BG>
struct BgCount {
  uint32_t wave;
  __device__ __forceinline__ void kmer3(uint32_t id, bool on) const {
    if (BG) atomicAdd(&bg_lds().bins[on ? (id >> (2 * W - 6)) : 95u], 1u);  // bin 95: sink
  }
  // 4-mer (first base in the low bits) ending on an owned base whose predecessor is owned too
  __device__ __forceinline__ void kmer4(uint32_t v) const {
    if (BG) atomicAdd(&bg_lds().bins4[v], 1u);
  }
  // base at run position sp (static) of a non-continuing item, id already rolled
  __device__ __forceinline__ void head(uint32_t id, int sp, bool on) const {
    if (!BG || !on || sp < 0) return;
    if (sp == 0) atomicAdd(&bg_lds().bins[64 + (id >> (2 * W - 2))], 1u);
    else if (sp == 1) atomicAdd(&bg_lds().bins[68 + (id >> (2 * W - 4))], 1u);
    else atomicAdd(&bg_lds().bins[id >> (2 * W - 6)], 1u);
  }
};

template <bool BG>
__device__ __forceinline__ void bg_begin() {
  if (BG) {
    for (uint32_t i = threadIdx.x; i < 96u; i += blockDim.x) bg_lds().bins[i] = 0;
    for (uint32_t i = threadIdx.x; i < 256u; i += blockDim.x) bg_lds().bins4[i] = 0;
  }
}
// block partials [gridDim.x][84]; summed in block order by bg_finish_fused_kernel (deterministic, no atomics)
template <bool BG>
__device__ __forceinline__ void bg_end(uint32_t* __restrict__ bg_partials) {
  if (BG) {
    __syncthreads();
    if (threadIdx.x < 84) {
      const BgLds& b = bg_lds();
      const uint32_t k = threadIdx.x;
      uint32_t v = b.bins[k];
      if (k < 64) {  // 4-mer v4 = b0 | b1 << 2 | b2 << 4 | b3 << 6 holds the 3-mers v4 & 63 and v4 >> 2
        for (uint32_t c = 0; c < 4; ++c) v += b.bins4[k | (c << 6)] + b.bins4[(k << 2) | c];
      }
      bg_partials[(size_t)blockIdx.x * 84 + k] = v;
    }
  }
}

// "can equals one of the ids of the last W-1 COUNTED windows" (src/base_pattern.cpp:361-366 restated, see the file
// header) as a chain of W-1 v_cmpx_ne_u32: every compare narrows EXEC to the lanes that have not matched yet, the
// final v_mov runs only in those, and EXEC is restored -- W+1 vector instructions instead of the 2(W-1) of an
// xor / min3 / compare / select chain (pass A of the partitioned count is bound by its VALU instruction count;
// v_cmp_eq into SGPR pairs + s_or_b64 saturated the CU's one scalar ALU instead, profiles/r01_v3_pmc_sq.txt).
// Returns can, or INVALID_ID for a suppressed window.  ring[(u - d) & 15] is the key of the window d positions back.
template <int W>
struct Suppress {
  static __device__ __forceinline__ uint32_t apply(uint32_t can, const uint32_t (&ring)[16], int u) {
    uint32_t out = INVALID_ID;
    unsigned long long save;
#define PENGK_R(d) "v"(ring[(u - (d)) & 15])
#define PENGK_CX(n) "v_cmpx_ne_u32_e32 %2, %" #n "\n\t"
#define PENGK_HEAD "s_mov_b64 %1, exec\n\t"
#define PENGK_TAIL "v_mov_b32_e32 %0, %2\n\ts_mov_b64 exec, %1"
    if constexpr (W == 2)
      asm volatile(PENGK_HEAD PENGK_CX(3) PENGK_TAIL : "+v"(out), "=&s"(save) : "v"(can), PENGK_R(1) : "vcc");
    else if constexpr (W == 4)
      asm volatile(PENGK_HEAD PENGK_CX(3) PENGK_CX(4) PENGK_CX(5) PENGK_TAIL
                   : "+v"(out), "=&s"(save) : "v"(can), PENGK_R(1), PENGK_R(2), PENGK_R(3) : "vcc");
    else if constexpr (W == 6)
      asm volatile(PENGK_HEAD PENGK_CX(3) PENGK_CX(4) PENGK_CX(5) PENGK_CX(6) PENGK_CX(7) PENGK_TAIL
                   : "+v"(out), "=&s"(save) : "v"(can), PENGK_R(1), PENGK_R(2), PENGK_R(3), PENGK_R(4), PENGK_R(5) : "vcc");
    else if constexpr (W == 8)
      asm volatile(PENGK_HEAD PENGK_CX(3) PENGK_CX(4) PENGK_CX(5) PENGK_CX(6) PENGK_CX(7) PENGK_CX(8) PENGK_CX(9) PENGK_TAIL
                   : "+v"(out), "=&s"(save)
                   : "v"(can), PENGK_R(1), PENGK_R(2), PENGK_R(3), PENGK_R(4), PENGK_R(5), PENGK_R(6), PENGK_R(7) : "vcc");
    else if constexpr (W == 10)
      asm volatile(PENGK_HEAD PENGK_CX(3) PENGK_CX(4) PENGK_CX(5) PENGK_CX(6) PENGK_CX(7) PENGK_CX(8) PENGK_CX(9) PENGK_CX(10)
                   PENGK_CX(11) PENGK_TAIL
                   : "+v"(out), "=&s"(save)
                   : "v"(can), PENGK_R(1), PENGK_R(2), PENGK_R(3), PENGK_R(4), PENGK_R(5), PENGK_R(6), PENGK_R(7), PENGK_R(8),
                     PENGK_R(9) : "vcc");
    else if constexpr (W == 12)
      asm volatile(PENGK_HEAD PENGK_CX(3) PENGK_CX(4) PENGK_CX(5) PENGK_CX(6) PENGK_CX(7) PENGK_CX(8) PENGK_CX(9) PENGK_CX(10)
                   PENGK_CX(11) PENGK_CX(12) PENGK_CX(13) PENGK_TAIL
                   : "+v"(out), "=&s"(save)
                   : "v"(can), PENGK_R(1), PENGK_R(2), PENGK_R(3), PENGK_R(4), PENGK_R(5), PENGK_R(6), PENGK_R(7), PENGK_R(8),
                     PENGK_R(9), PENGK_R(10), PENGK_R(11) : "vcc");
    else {
      static_assert(W == 14, "pattern lengths 2 .. 14");
      asm volatile(PENGK_HEAD PENGK_CX(3) PENGK_CX(4) PENGK_CX(5) PENGK_CX(6) PENGK_CX(7) PENGK_CX(8) PENGK_CX(9) PENGK_CX(10)
                   PENGK_CX(11) PENGK_CX(12) PENGK_CX(13) PENGK_CX(14) PENGK_CX(15) PENGK_TAIL
                   : "+v"(out), "=&s"(save)
                   : "v"(can), PENGK_R(1), PENGK_R(2), PENGK_R(3), PENGK_R(4), PENGK_R(5), PENGK_R(6), PENGK_R(7), PENGK_R(8),
                     PENGK_R(9), PENGK_R(10), PENGK_R(11), PENGK_R(12), PENGK_R(13) : "vcc");
    }
#undef PENGK_R
#undef PENGK_CX
#undef PENGK_HEAD
#undef PENGK_TAIL
    return out;
  }
};

template <int W, bool BOTH, bool BG, class Emit>
__device__ __forceinline__ void scan_items(const uint32_t* __restrict__ words32, const uint64_t* __restrict__ items,
                                           uint32_t n_items, unsigned long long* __restrict__ ltot,
                                           uint32_t* __restrict__ defer, Emit& emit) {
  using G = Geo<W>;
  const BgCount<W, BG> bgc{threadIdx.x >> 6};
  const uint32_t lane_global = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t stride = gridDim.x * blockDim.x;
  unsigned long long my_windows = 0;

  // wave-uniform trip count: every lane of a wave runs the same number of outer iterations
  for (uint32_t wave_base = lane_global & ~63u; wave_base < n_items; wave_base += stride) {
    const uint32_t it = wave_base + (threadIdx.x & 63u);
    const bool live = it < n_items;
    const uint64_t rec = live ? items[it] : 0ull;
    uint32_t nw = (uint32_t)((rec >> ITEM_NW_SHIFT) & ITEM_NW_MASK);
    const uint64_t ws = rec & ITEM_WS_MASK;
    const bool cont = ((rec >> ITEM_CONT_SHIFT) & 1ull) != 0;
    const uint32_t nw_all = nw;  // a deferred item (nw = 0 below) still owns its bases for the fused K1b
    my_windows += nw;

    uint32_t ring[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) ring[i] = INVALID_ID;
    uint32_t id = 0, rc = 0;

    // local base 0 of the prologue = global base ws + (W-1) - P (>= 0 thanks to the front pad)
    const uint64_t g0 = live ? ws + (uint64_t)(W - 1) - (uint64_t)G::P : 0ull;
    uint64_t wi = g0 >> 4;
    const uint32_t shift = 2u * (uint32_t)(g0 & 15u);
    uint32_t lo;
    uint32_t pchunk = 0;  // the 16 bases in front of base P, aligned (last chunk of either prologue)

    if (__any(cont ? 1 : 0)) {
      // full prologue: rebuild the ring of counted windows in front of a continuing item
      lo = words32[wi];
      bool dirty = false;
#pragma unroll
      for (int ch = 0; ch < G::P / 16; ++ch) {
        const uint32_t hi = words32[wi + 1];
        ++wi;
        const uint32_t chunk = funnel(hi, lo, shift);
        lo = hi;
        pchunk = chunk;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int b = ch * 16 + u;
          const uint32_t c = (chunk >> (2 * u)) & 3u;
          id = (id >> 2) | (c << G::TOP);
          rc = ((rc << 2) & G::MASK) | (c ^ 3u);
          bgc.head(id, b - (G::P - (W - 1)), live && !cont && nw > 0);
          if (b >= W - 1) {
            const uint32_t can = BOTH ? min(id, rc) : id;
            bool match = false;
#pragma unroll
            for (int d = 1; d <= W - 1; ++d)
              if (b - d >= W - 1) match |= (can == ring[(u - d) & 15]);
            dirty |= match;
            ring[u] = (cont && !match) ? can : INVALID_ID;
          }
        }
      }
      if (cont && dirty) {  // cannot certify the ring: hand the item to the exact fallback
        const uint32_t slot = atomicAdd(&defer[0], 1u);
        defer[1 + slot] = it;
        nw = 0;
      }
    } else {
      // no lane continues a run: only the last W-1 bases in front of base P matter (id / rc state)
      wi += (G::P / 16) - 1;
      lo = words32[wi];
      const uint32_t hi = words32[wi + 1];
      ++wi;
      const uint32_t chunk = funnel(hi, lo, shift);
      lo = hi;
      pchunk = chunk;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const uint32_t c = (chunk >> (2 * u)) & 3u;
        id = (id >> 2) | (c << G::TOP);
        rc = ((rc << 2) & G::MASK) | (c ^ 3u);
        bgc.head(id, u - (17 - W), live && nw > 0);
      }
    }

    // main scan: window t ends at local base P + t.  The loop is wave-uniform (longest item of the
    // wave) so that emitters may use wave-wide operations.
    const uint32_t nw_scan = BG ? nw_all : nw;
    uint32_t nw_max = nw_scan;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nw_max = max(nw_max, (uint32_t)__shfl_xor((int)nw_max, off, 64));
    // the stream word of the NEXT iteration is requested one iteration ahead: vmcnt retires in order, so a
    // load issued after this iteration's key stores (scatter variant) would wait for all of them
    uint32_t nxt = (0u < nw_scan) ? words32[wi + 1] : 0u;
    // In the main scan ids are not rolled base by base: the stream is little-endian like the ids, so the id of
    // the window that ends on base u of this chunk is the bit field [2(u+17-W), +2W) of chunk:pchunk, and its
    // reverse complement the field [2(15-u), +2W) of revcomp(chunk):revcomp(pchunk) -- one v_bfe each, from the
    // words themselves or from a few 32-bit views of the straddling part (7 VALU per window -> 3.6; pass A of
    // the partitioned count is bound by its VALU instruction count).
    uint32_t rprev = revcomp16(pchunk);
    for (uint32_t t0 = 0; t0 < nw_max; t0 += 16) {
      const uint32_t hi = nxt;
      ++wi;
      nxt = (t0 + 16u < nw_scan) ? words32[wi + 1] : 0u;
      const uint32_t chunk = funnel(hi, lo, shift);
      lo = hi;
      const uint32_t rchunk = BOTH ? revcomp16(chunk) : 0u;
      uint32_t view[G::NVIEW], rview[G::NVIEW];
#pragma unroll
      for (int j = 0; j < G::NVIEW; ++j) {
        view[j] = funnel(chunk, pchunk, 2u * (uint32_t)(j * G::PER + 17 - W));
        rview[j] = BOTH ? funnel(rprev, rchunk, 2u * (uint32_t)(j * G::PER + 17 - W)) : 0u;
      }
      // Two bodies.  FULL (wave-uniform: every lane owns all 16 windows of this chunk -- all chunks but the last of
      // equally long items): no per-window range tests, suppressed windows travel as INVALID_ID keys, and the fused
      // K1b counts one 4-mer per two bases.  Otherwise the per-window predicates decide.
      auto body = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          uint32_t id_u, rc_u = 0;
          if (u >= W - 1) {  // the window lies inside this chunk
            id_u = __builtin_amdgcn_ubfe(chunk, 2u * (uint32_t)(u - (W - 1)), 2u * W);
            if (BOTH) rc_u = __builtin_amdgcn_ubfe(rchunk, 2u * (uint32_t)(15 - u), 2u * W);
          } else {
            const int j = u / G::PER, k = W - 2 - u, jr = k / G::PER;
            id_u = __builtin_amdgcn_ubfe(view[j], 2u * (uint32_t)(u - j * G::PER), 2u * W);
            if (BOTH) rc_u = __builtin_amdgcn_ubfe(rview[jr], 2u * (uint32_t)(k - jr * G::PER), 2u * W);
          }
          const uint32_t can = BOTH ? min(id_u, rc_u) : id_u;
#ifdef PENGK_ABLATE_NOSUPPRESS  // timing experiment only (WRONG counts on repeats): what a free repeat pre-filter could gain at most
          const uint32_t key = can;
#else
          const uint32_t key = Suppress<W>::apply(can, ring, u);  // INVALID_ID iff one of the last W-1 counted ids
#endif
          ring[u] = key;
          if (FULL) {
            if (BG && (u & 1)) {  // 4-mer ending on base u: bases u-3 .. u
              const uint32_t v4 = u >= 3 ? __builtin_amdgcn_ubfe(chunk, 2u * (uint32_t)(u - 3), 8u)
                                         : funnel(chunk, pchunk, 2u * (uint32_t)(13 + u)) & 0xFFu;
              bgc.kmer4(v4);
            }
            emit.full(key);
          } else {
            bgc.kmer3(id_u, t0 + (uint32_t)u < nw_all);
            emit.masked(key, key != INVALID_ID && t0 + (uint32_t)u < nw);
          }
        }
      };
      if (__all((t0 + 16u <= nw && nw == nw_all) ? 1 : 0)) body(std::true_type{});
      else body(std::false_type{});
      pchunk = chunk;
      rprev = rchunk;
    }
  }

  // ltot: one atomic per wave
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) my_windows += __shfl_down(my_windows, off, 64);
  if ((threadIdx.x & 63u) == 0 && my_windows) atomicAdd(ltot, my_windows);
}

// ---------------------------------------------------------------------------------------------
// K1 variant 1 ("direct"): one device-scope atomic per counted window.  Runs at the chip's
// scattered-atomic rate (~2.5e10/s measured); kept for W = 4, 6, 12, 14 and as the A/B baseline.
// hist: uint32[4^W]; ltot: uint64 scalar; defer[0] = number of deferred items, defer[1..] indices.
// ---------------------------------------------------------------------------------------------
struct DirectEmit {
  uint32_t* __restrict__ hist;
  __device__ __forceinline__ void masked(uint32_t key, bool active) const {
    if (active) __hip_atomic_fetch_add(&hist[key], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ __forceinline__ void full(uint32_t key) const { masked(key, key != INVALID_ID); }
};

template <int W, bool BOTH, bool BG>
__global__ __launch_bounds__(256) void count_kernel(const uint32_t* __restrict__ words32,
                                                    const uint64_t* __restrict__ items, uint32_t n_items,
                                                    uint32_t* __restrict__ hist,
                                                    unsigned long long* __restrict__ ltot,
                                                    uint32_t* __restrict__ defer, uint32_t* __restrict__ bg_partials) {
  bg_begin<BG>();
  if (BG) __syncthreads();
  DirectEmit e{hist};
  scan_items<W, BOTH, BG>(words32, items, n_items, ltot, defer, e);
  bg_end<BG>(bg_partials);
}

// ---------------------------------------------------------------------------------------------
// K1 variant 2 ("partition"): LDS-privatised histograms for 4^W bins that do not fit in LDS.
//
//   pass A  count_scatter_kernel: the same scan, but a counted window's id is split into a bucket
//           (NBITS bits from the middle of the id, see KeySplit) and a 15-bit payload.  Every wave owns NB small
//           rings in LDS (128 x u16 each); whenever a ring completes a group of 64 payloads the
//           wave writes them as ONE 128-byte line into ITS OWN slice of the bucket's region of a
//           key buffer in HBM.  The slices are static (region[wave][bucket][slice_cap]): no
//           reservation atomics -- a first version reserved chunks from 32 shared cursors and spent
//           8 of its 12 ms queueing on those 32 addresses.  2 B of HBM write per counted window.
//   pass B  count_hist_kernel: a workgroup owns part of one bucket, keeps the bucket's 2^15 bins in
//           128 KiB of LDS, streams the slices (16 B per lane) and counts with LDS atomics; the
//           block histogram is added to a bucket-major table with coalesced atomics.
//   pass C  count_gather_kernel: table[join(bucket, payload)] += bucket-major table.
//
// Skewed inputs cannot break it: a slice that runs full makes further groups of that (wave, bucket)
// fall back to direct atomics on the final table (graceful degradation to variant 1).
// ---------------------------------------------------------------------------------------------
constexpr int RING_CAP = 128;           // u16 entries per (wave, bucket) ring: a group of 64 plus 64 in flight
constexpr int GROUP = RING_CAP / 2;        // entries written per flush (one 128-byte line at RING_CAP = 128)
constexpr uint32_t KEY_INVALID = 0xFFFFu;
constexpr int PAYLOAD_BITS = 15;

// Bucket = NBITS bits from the MIDDLE of the id, payload = the remaining 15 bits squeezed together.
// The low and high digits must not choose the bucket: a canonical id satisfies id <= revcomp(id), which
// couples its first and last digits (P(first digit = A,C,G,T) = .4,.3,.2,.1), and low-bit buckets were
// loaded 1.6 : 0.4 -- every tenth key overflowed its slice into the direct-atomic path.  The middle
// digits are compared last by that inequality and stay uniform.
template <int W, int NBITS>
struct KeySplit {
  static constexpr int S = ((2 * W - NBITS) / 2) & ~1;  // first bucket bit
  static constexpr uint32_t LOW = (1u << S) - 1u;
  static constexpr uint32_t NB = 1u << NBITS;
  __host__ __device__ static inline uint32_t bucket(uint32_t id) { return (id >> S) & (NB - 1u); }
  // = (id & LOW) | ((id >> (S + NBITS)) << S), as a bit-field insert: one shift + v_bfi_b32 (the compiler splits
  // the portable expression into three instructions, and pass A is bound by its VALU instruction count)
  __host__ __device__ static inline uint32_t payload(uint32_t id) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(LOW), "v"(id), "v"(id >> NBITS));
    return r;
#else
    return (id & LOW) | ((id >> NBITS) & ~LOW);
#endif
  }
  __host__ __device__ static inline uint32_t join(uint32_t b, uint32_t p, uint32_t /*outer*/ = 0) {
    return (p & LOW) | (b << S) | ((p >> S) << (S + NBITS));
  }
  __host__ __device__ static inline bool valid(uint32_t p) { return p != KEY_INVALID; }
};

// W = 12 (24-bit ids) needs 2^9 buckets of 2^15 bins: two levels.
//   level 1: bucket1 = id bits [8,13), 19-bit payload1 = the rest squeezed together (32-bit keys)
//   level 2: bucket2 = payload1 bits [8,12), 15-bit payload2 = the rest (16-bit keys)
// Again the bucket bits come from the middle of the id (see KeySplit).
#ifndef PENGK_W12_L1_BITS
#define PENGK_W12_L1_BITS 4  // bucket bits of level 1 (level 2 takes 9 - that).  Measured on a 12.5M x 200 bp shard: 5 + 4 -> 8.26 ms,
                             // 4 + 5 -> 7.2 (the scan's 32-bit rings are 8.3 instead of 16.5 KiB per wave: 16 instead of 9
                             // waves per CU for the kernel that is bound by what it issues), 3 + 6 -> 8.86
#endif
struct Split12L1 {
  static constexpr int NBITS = PENGK_W12_L1_BITS;
  static constexpr uint32_t NB = 1u << NBITS;
  static constexpr uint32_t PBITS = 24 - NBITS;  // bits of payload1
  __host__ __device__ static inline uint32_t bucket(uint32_t id) { return (id >> 8) & (NB - 1u); }
  __host__ __device__ static inline uint32_t payload(uint32_t id) { return (id & 0xFFu) | ((id >> (8 + NBITS)) << 8); }
  __host__ __device__ static inline uint32_t join(uint32_t b1, uint32_t p1, uint32_t /*outer*/ = 0) { return (p1 & 0xFFu) | (b1 << 8) | ((p1 >> 8) << (8 + NBITS)); }
  // a payload of PBITS bits; INVALID_ID and padding entries carry higher bits (and travel on as KEY_INVALID through level 2)
  __host__ __device__ static inline bool valid(uint32_t p1) { return p1 < (1u << PBITS); }
};
struct Split12L2 {
  static constexpr int NBITS = 9 - Split12L1::NBITS;
  static constexpr uint32_t NB = 1u << NBITS;
  __host__ __device__ static inline uint32_t bucket(uint32_t p1) { return (p1 >> 8) & (NB - 1u); }
  __host__ __device__ static inline uint32_t payload(uint32_t p1) { return (p1 & 0xFFu) | ((p1 >> (8 + NBITS)) << 8); }
  // (bucket2, payload2) of level-1 bucket `outer` -> 24-bit id
  __host__ __device__ static inline uint32_t join(uint32_t b2, uint32_t p2, uint32_t outer) {
    const uint32_t p1 = (p2 & 0xFFu) | (b2 << 8) | ((p2 >> 8) << (8 + NBITS));
    return Split12L1::join(outer, p1);
  }
  __host__ __device__ static inline bool valid(uint32_t p2) { return p2 != KEY_INVALID; }
};

// W = 14 (28-bit ids) needs 2^13 buckets of 2^15 bins: three levels, bucket bits again from the middle of what is left.
//   level 1: bucket1 = id bits [10,14), 24-bit payload1 (32-bit keys; 16 rings of 32-bit entries per wave, like W = 12's)
//   level 2: bucket2 = payload1 bits [9,13), 20-bit payload2 (32-bit keys)
//   level 3: bucket3 = payload2 bits [8,13), 15-bit payload3 (16-bit keys)
// A suppressed window (INVALID_ID) and the padding entries are all ones in every payload bit at every level: their
// bucket bits select the last bucket, their payloads fail valid() and end as KEY_INVALID, which pass B skips.
struct Split14L1 {
  static constexpr int NBITS = 4;
  static constexpr uint32_t NB = 16;
  __host__ __device__ static inline uint32_t bucket(uint32_t id) { return (id >> 10) & 15u; }
  __host__ __device__ static inline uint32_t payload(uint32_t id) { return (id & 0x3FFu) | ((id >> 14) << 10); }
  __host__ __device__ static inline uint32_t join(uint32_t b1, uint32_t p1, uint32_t /*outer*/ = 0) { return (p1 & 0x3FFu) | (b1 << 10) | ((p1 >> 10) << 14); }
  __host__ __device__ static inline bool valid(uint32_t p1) { return p1 < (1u << 24); }
};
struct Split14L2 {
  static constexpr int NBITS = 4;
  static constexpr uint32_t NB = 16;
  __host__ __device__ static inline uint32_t bucket(uint32_t p1) { return (p1 >> 9) & 15u; }
  __host__ __device__ static inline uint32_t payload(uint32_t p1) { return (p1 & 0x1FFu) | ((p1 >> 13) << 9); }
  // (bucket2, payload2) of level-1 bucket `outer` -> 28-bit id
  __host__ __device__ static inline uint32_t join(uint32_t b2, uint32_t p2, uint32_t outer) {
    return Split14L1::join(outer, (p2 & 0x1FFu) | (b2 << 9) | ((p2 >> 9) << 13));
  }
  __host__ __device__ static inline bool valid(uint32_t p2) { return p2 < (1u << 20); }
};
struct Split14L3 {
  static constexpr int NBITS = 5;
  static constexpr uint32_t NB = 32;
  __host__ __device__ static inline uint32_t bucket(uint32_t p2) { return (p2 >> 8) & 31u; }
  __host__ __device__ static inline uint32_t payload(uint32_t p2) { return (p2 & 0xFFu) | ((p2 >> 13) << 8); }
  // (bucket3, payload3) of the level-1 / level-2 buckets outer = b1 * 16 + b2 -> 28-bit id
  __host__ __device__ static inline uint32_t join(uint32_t b3, uint32_t p3, uint32_t outer) {
    return Split14L2::join(outer & 15u, (p3 & 0xFFu) | (b3 << 8) | ((p3 >> 8) << 13), outer >> 4);
  }
  __host__ __device__ static inline bool valid(uint32_t p3) { return p3 != KEY_INVALID; }
};

// One row per (wave, bucket): the ring and, behind it, its counter.  The odd row stride (65 dwords) spreads rows over
// the LDS banks: rings fill at the same pace, and with a 256-byte stride equal fill levels would put every lane of
// the 16-bit ring write -- and every counter -- on the same few banks.
template <class E>  // E = uint16_t (15-bit payloads) or uint32_t (the 19-bit payloads of W = 12's first level)
struct ScatterRowT {
  E ring[RING_CAP];
  uint32_t fill4;  // 4 x (keys ever appended to this (wave, bucket))
};
typedef ScatterRowT<uint16_t> ScatterRow;
template <int NBITS, int WPW, class E = uint16_t>
struct ScatterShared {
  static constexpr int NB = 1 << NBITS;
  ScatterRowT<E> row[WPW][NB];
};

// The one LDS instance per workgroup.  It is reached through this accessor, never through a pointer
// stored in a struct: a generic pointer made the compiler emit flat_load for the ring reads, and a flat
// access waits for vmcnt(0) -- i.e. for every key store still in flight -- on every flush.
template <int NBITS, int WPW, class E = uint16_t>
__device__ __forceinline__ ScatterShared<NBITS, WPW, E>& scatter_lds() {
  __shared__ ScatterShared<NBITS, WPW, E> sh;
  return sh;
}

// No cursor state at all: a row's counter is an LDS atomic that counts every key ever appended to the (wave, bucket),
// so the value a lane gets back IS the key's position in the slice -- the group that completes with slot s goes to
// slice entries [s - 63, s], an address formed from wave-uniform scalars (SALU) plus lane * 2.  Ring payloads written
// by other lanes are read through a volatile LDS pointer.  (Earlier versions kept a write cursor per bucket in LDS
// words -- the compiler legally re-used stale copies -- and then in the registers of lane b: three v_readlane, a
// 64-bit add and two selects per flush.)
//
// The counter advances by 4 per key: byte 0 of the returned value is then 4 * (slot mod 64), so "this key completes
// a group" is ONE sub-dword compare, and (value >> 1) & 0xFF is the byte offset of the key's ring entry.  A
// suppressed window arrives as INVALID_ID: its bucket bits select the last bucket and its payload bits are
// KEY_INVALID, which pass B skips -- no select, no sink row (7 vector instructions per key instead of 14).
template <class KS, int NBITS, int WPW = 4, class E = uint16_t>
struct ScatterEmit {
  static constexpr int NB = 1 << NBITS;
  static constexpr uint32_t ES = (uint32_t)sizeof(E);           // bytes per ring / slice entry
  static constexpr uint32_t ESH = ES == 2u ? 1u : 2u;            // log2(ES)
  static constexpr uint32_t PAD = ES == 2u ? KEY_INVALID : 0xFFFFFFFFu;  // an entry pass B / the next level skips
  typedef ScatterRowT<E> Row;
  static_assert(GROUP == 64, "a group is one entry per lane");
  static_assert(RING_CAP == 128, "byte 0 of the counter addresses the ring");
  E* __restrict__ keys;
  uint32_t slice_cap;  // entries per (wave, bucket) slice, multiple of 64; NB * slice_cap * ES < 2^32
  uint32_t* __restrict__ slice_fill;  // [n_waves][NB] entries written (multiple of 64)
  uint32_t* __restrict__ hist;
  uint32_t wave, lane, wave_global;   // wave, wave_global: wave-uniform (readfirstlane'd by the caller)
  uint32_t outer;     // level-1 bucket these keys came from (two-level partition); 0 otherwise

  // All threads of the workgroup, before the first barrier: rings all ones, counters 0.
  static __device__ __forceinline__ void init_lds() {
    ScatterShared<NBITS, WPW, E>& sh = scatter_lds<NBITS, WPW, E>();
    static_assert(sizeof(Row) == ES * RING_CAP + 4, "row = ring + counter");
    uint32_t* w = reinterpret_cast<uint32_t*>(&sh.row[0][0]);
    for (uint32_t i = threadIdx.x; i < (uint32_t)WPW * NB * (sizeof(Row) / 4u); i += blockDim.x)
      w[i] = (i % (uint32_t)(sizeof(Row) / 4u)) == (uint32_t)(ES * RING_CAP / 4u) ? 0u : 0xFFFFFFFFu;
  }

  typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
  typedef __attribute__((address_space(3))) E lds_e_t;
  typedef __attribute__((address_space(1))) E global_e_t;

  // A completed group (64 ring entries from slot g0, a multiple of 64) goes to slice entries [g0, g0 + 64) as ONE
  // line (128 bytes of 16-bit entries, 256 of 32-bit ones), one load and store per lane.  Everything wave-uniform about
  // a group -- where it starts in the ring, where it goes in the wave's key region, whether the slice still has room --
  // is computed by the lane whose key completed it, with a handful of vector instructions that all lanes run side by
  // side; the loop then needs two v_readlane per group and no scalar multiply / shift / mask chain (13 instruction
  // issues per group instead of 30; the flush was half of what this kernel issues).
  __device__ __forceinline__ void flush_triggered(unsigned long long trig, uint32_t row, uint32_t b, uint32_t s4) {
    // (an empty asm: the compiler must not fold the caller's wave-uniform "some key completed a group" branch into the
    // masks below -- that turns one scalar branch per key into five scalar instructions)
    asm volatile("");
    const uint32_t g4 = s4 - 4u * (uint32_t)(GROUP - 1);        // 4 x first slot of the group (in trigger lanes)
    const uint32_t src = row + ((g4 >> (2u - ESH)) & (ES * (uint32_t)GROUP));  // ring byte offset: first or second half
    const uint32_t dst = b * (ES * slice_cap) + (g4 >> (2u - ESH));  // byte offset in the wave's region
    const unsigned long long fits = __builtin_amdgcn_ballot_w64(g4 + 4u * GROUP <= 4u * slice_cap);
    unsigned long long todo = trig & fits;
    unsigned long long over = trig & ~fits;
    while (todo) {  // wave-uniform: on average one group per step
      const int p = __builtin_ctzll(todo);
      todo &= todo - 1;
      const uint32_t s_src = (uint32_t)__builtin_amdgcn_readlane((int)src, p);
      const uint32_t s_dst = (uint32_t)__builtin_amdgcn_readlane((int)dst, p);
      // A plain LDS load (a volatile one makes the compiler drain lgkmcnt(0) in front of it).  The LDS executes a wave's
      // operations in order, so it sees every ring write issued before it; ring writes through integer-formed addresses
      // lie between two reads of a location, so it cannot be satisfied from an older copy.
      uint32_t v = *(lds_e_t*)(uintptr_t)(s_src + ES * lane);
      asm volatile("" : "+v"(v));  // consumed here on every path: no wait for it leaks into the next window's code
      global_e_t* out = (global_e_t*)((char*)region + s_dst);
      __builtin_nontemporal_store((E)v, &out[lane]);  // written once, read once by the next pass
    }
    while (over) {  // slice full: count these windows directly (rare; skewed inputs)
      const int p = __builtin_ctzll(over);
      over &= over - 1;
      const uint32_t s_src = (uint32_t)__builtin_amdgcn_readlane((int)src, p);
      const uint32_t fb = (uint32_t)__builtin_amdgcn_readlane((int)b, p);
      const uint32_t v = *(lds_e_t*)(uintptr_t)(s_src + ES * lane);
      if (KS::valid(v)) __hip_atomic_fetch_add(&hist[KS::join(fb, v, outer)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }

  // LDS byte address of this wave's row 0, held in a VECTOR register on purpose (opaque to the compiler): the row
  // address is then one v_mad_u32_u24 (a VALU instruction reads at most one scalar operand on gfx950, and the
  // compiler otherwise re-materialises the scalar base with a v_mov per key).
  uint32_t rowbase = 0;
  __device__ __forceinline__ void bind() {
    rowbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)&scatter_lds<NBITS, WPW, E>().row[wave][0];
    asm volatile("" : "+v"(rowbase));
    region = keys + (size_t)wave_global * NB * slice_cap;
  }
  E* region = nullptr;  // this wave's key region (wave-uniform): NB slices of slice_cap entries

  // append one key per lane: the row of its bucket (returned through `row`) and 4 x its slot
  __device__ __forceinline__ uint32_t append(uint32_t b, uint32_t key, uint32_t& row) const {
    row = b * (uint32_t)sizeof(Row) + rowbase;  // b < NB: one v_mad_u32_u24
    const uint32_t s4 = __hip_atomic_fetch_add((lds_u32_t*)(uintptr_t)(row + ES * RING_CAP), 4u, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP);
    *(lds_e_t*)(uintptr_t)(row + ((s4 >> (2u - ESH)) & (ES * RING_CAP - 1u))) = (E)KS::payload(key);
    return s4;
  }

  // every lane appends; key may be INVALID_ID (or any value whose payload bits are all ones)
  __device__ __forceinline__ void full(uint32_t key) {
    const uint32_t b = KS::bucket(key);
    uint32_t row;
    const uint32_t s4 = append(b, key, row);
    const unsigned long long trig = __builtin_amdgcn_ballot_w64((s4 & 0xFFu) == 4u * (uint32_t)(GROUP - 1));
    if (trig) flush_triggered(trig, row, b, s4);  // wave-uniform
  }

  // only the lanes with `active` append
  __device__ __forceinline__ void masked(uint32_t key, bool active) {
    const uint32_t b = KS::bucket(key);
    uint32_t s4 = 0, row = 0;
    if (active) s4 = append(b, key, row);
    const unsigned long long trig = __builtin_amdgcn_ballot_w64(active && (s4 & 0xFFu) == 4u * (uint32_t)(GROUP - 1));
    if (trig) flush_triggered(trig, row, b, s4);
  }

  // end of kernel: lane b pads bucket b's last, partial group with all-ones entries -- appending the padding, so that
  // the group is complete and leaves like every other -- and publishes how much of the slice is filled
  __device__ __forceinline__ void drain() {
    __builtin_amdgcn_wave_barrier();
    uint32_t row = 0, s4_last = 0;
    bool padded = false;
    if (lane < (uint32_t)NB) {
      row = lane * (uint32_t)sizeof(Row) + rowbase;
      const uint32_t c4 = __hip_atomic_load((lds_u32_t*)(uintptr_t)(row + ES * RING_CAP), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      const uint32_t r = (c4 >> 2) & (uint32_t)(GROUP - 1);
      if (r) {
        for (uint32_t i = r; i < (uint32_t)GROUP; ++i)
          *(lds_e_t*)(uintptr_t)(row + (((c4 >> (2u - ESH)) & (ES * RING_CAP - 1u)) + ES * (i - r))) = (E)PAD;
        s4_last = c4 + 4u * ((uint32_t)GROUP - r) - 4u;  // the slot value the completing key would have got
        padded = true;
      }
      const uint32_t full = ((c4 >> 2) + (uint32_t)(GROUP - 1)) & ~(uint32_t)(GROUP - 1);
      slice_fill[(size_t)wave_global * NB + lane] = full < slice_cap ? full : slice_cap;
    }
    __builtin_amdgcn_wave_barrier();
    const unsigned long long trig = __builtin_amdgcn_ballot_w64(padded);
    if (trig) flush_triggered(trig, row, lane, s4_last);
  }
};

#ifndef PENGK_SCATTER_WPW
#define PENGK_SCATTER_WPW 4
#endif
constexpr int SCATTER_WPW = PENGK_SCATTER_WPW;  // waves per workgroup of pass A
template <int W, bool BOTH, int NBITS, bool BG>
__global__ __launch_bounds__(64 * SCATTER_WPW) void count_scatter_kernel(const uint32_t* __restrict__ words32,
                                                            const uint64_t* __restrict__ items, uint32_t n_items,
                                                            uint16_t* __restrict__ keys, uint32_t slice_cap,
                                                            uint32_t* __restrict__ slice_fill, uint32_t* __restrict__ hist,
                                                            unsigned long long* __restrict__ ltot,
                                                            uint32_t* __restrict__ defer,
                                                            uint32_t* __restrict__ bg_partials) {
  static_assert(2 * W - NBITS == PAYLOAD_BITS, "payload must be 15 bits");
  ScatterEmit<KeySplit<W, NBITS>, NBITS, SCATTER_WPW>::init_lds();
  bg_begin<BG>();
  __syncthreads();
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  ScatterEmit<KeySplit<W, NBITS>, NBITS, SCATTER_WPW> e{keys, slice_cap, slice_fill, hist, wave, threadIdx.x & 63u, blockIdx.x * (uint32_t)SCATTER_WPW + wave, 0u};
  e.bind();
  scan_items<W, BOTH, BG>(words32, items, n_items, ltot, defer, e);
  e.drain();
  bg_end<BG>(bg_partials);
}

// ---- two-level partition (W = 12) -----------------------------------------------------------------------
// Level 1: same scan, same emitter with 32-bit ring entries (19-bit payload1): 32 wave-private rings of 128 entries,
// a group of 64 leaves as one 256-byte line into region1[wave][bucket1][cap1].  16.5 KiB of rings per wave: three
// waves per workgroup, three workgroups per CU.  (Round 1's level 1 appended in two half-waves per window into
// 64-entry rings, with range tests and a sink bucket: 6.1 ms for a 12.5M-sequence shard.)
#ifndef PENGK_SCATTER12_WPW
#define PENGK_SCATTER12_WPW 3
#endif
constexpr int SCATTER12_WPW = PENGK_SCATTER12_WPW;
#ifndef PENGK_SCATTER12L1_WPW
#define PENGK_SCATTER12L1_WPW (PENGK_W12_L1_BITS == 5 ? 3 : 4)
#endif
constexpr int SCATTER12L1_WPW = PENGK_SCATTER12L1_WPW;  // waves per workgroup of W = 12's scan
typedef ScatterEmit<Split12L1, Split12L1::NBITS, SCATTER12L1_WPW, uint32_t> Scatter12Emit;

template <bool BOTH, bool BG>
__global__ __launch_bounds__(64 * SCATTER12L1_WPW) void count_scatter12_kernel(const uint32_t* __restrict__ words32,
                                                              const uint64_t* __restrict__ items, uint32_t n_items,
                                                              uint32_t* __restrict__ keys, uint32_t slice_cap,
                                                              uint32_t* __restrict__ slice_fill, uint32_t* __restrict__ hist,
                                                              unsigned long long* __restrict__ ltot,
                                                              uint32_t* __restrict__ defer, uint32_t* __restrict__ bg_partials) {
  Scatter12Emit::init_lds();
  bg_begin<BG>();
  __syncthreads();
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  Scatter12Emit e{keys, slice_cap, slice_fill, hist, wave, threadIdx.x & 63u, blockIdx.x * (uint32_t)SCATTER12L1_WPW + wave, 0u};
  e.bind();
  scan_items<12, BOTH, BG>(words32, items, n_items, ltot, defer, e);
  e.drain();
  bg_end<BG>(bg_partials);
}

// Level 2: a workgroup belongs to ONE level-1 bucket (blockIdx.x / bpb1); its waves stream that bucket's
// level-1 slices (64 keys per step, one per lane) and re-scatter them by bucket2 into 16 wave-private rings
// -> region2[wave2][bucket2][cap2] (16-bit payload2), exactly like pass A of the one-level scheme.
__global__ __launch_bounds__(256) void count_rescatter12_kernel(const uint32_t* __restrict__ keys1, uint32_t cap1,
                                                                const uint32_t* __restrict__ fill1, uint32_t n_slices1,
                                                                uint32_t bpb1, uint16_t* __restrict__ keys2, uint32_t cap2,
                                                                uint32_t* __restrict__ fill2, uint32_t* __restrict__ hist) {
  ScatterEmit<Split12L2, Split12L2::NBITS>::init_lds();
  __syncthreads();
  const uint32_t b1 = blockIdx.x / bpb1, j = blockIdx.x % bpb1;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
  ScatterEmit<Split12L2, Split12L2::NBITS> e{keys2, cap2, fill2, hist, wave, lane, blockIdx.x * 4u + wave, b1};
  e.bind();
  const uint32_t per = (n_slices1 + bpb1 - 1) / bpb1;
  const uint32_t first = j * per, last = min(n_slices1, first + per);
  for (uint32_t s = first + wave; s < last; s += 4) {
    const uint32_t n = fill1[(size_t)s * Split12L1::NB + b1];  // multiple of GROUP
    const uint32_t* src = keys1 + ((size_t)s * Split12L1::NB + b1) * cap1;
    // 256 keys per step: one 16-byte load per lane, the next step's already in flight while these four are appended
    // (one key per lane and step left the wave waiting for every line it asked for).  Padding and suppressed-window
    // entries (bits above the 19-bit payload set) need no test: their bucket bits select the last bucket and their
    // payload bits are KEY_INVALID, which pass B skips.
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 pad = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    const u4* src4 = reinterpret_cast<const u4*>(src);  // slices start on multiples of cap1 (a multiple of 32 entries)
    const uint32_t n4 = n / 4u;                         // n is a multiple of GROUP
    u4 nxt = lane < n4 ? __builtin_nontemporal_load(&src4[lane]) : pad;
    for (uint32_t i = 0; i < n4; i += 64) {  // wave-uniform trip count
      const u4 k = nxt;
      nxt = (i + 64u + lane < n4) ? __builtin_nontemporal_load(&src4[i + 64u + lane]) : pad;
      e.full(k.x);
      e.full(k.y);
      e.full(k.z);
      e.full(k.w);
    }
  }
  e.drain();
}

// ---- three-level partition (W = 14) ------------------------------------------------------------------------
typedef ScatterEmit<Split14L1, Split14L1::NBITS, SCATTER12L1_WPW, uint32_t> Scatter14Emit;

template <bool BOTH, bool BG>
__global__ __launch_bounds__(64 * SCATTER12L1_WPW) void count_scatter14_kernel(const uint32_t* __restrict__ words32,
                                                              const uint64_t* __restrict__ items, uint32_t n_items,
                                                              uint32_t* __restrict__ keys, uint32_t slice_cap,
                                                              uint32_t* __restrict__ slice_fill, uint32_t* __restrict__ hist,
                                                              unsigned long long* __restrict__ ltot,
                                                              uint32_t* __restrict__ defer, uint32_t* __restrict__ bg_partials) {
  Scatter14Emit::init_lds();
  bg_begin<BG>();
  __syncthreads();
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  Scatter14Emit e{keys, slice_cap, slice_fill, hist, wave, threadIdx.x & 63u, blockIdx.x * (uint32_t)SCATTER12L1_WPW + wave, 0u};
  e.bind();
  scan_items<14, BOTH, BG>(words32, items, n_items, ltot, defer, e);
  e.drain();
  bg_end<BG>(bg_partials);
}

// A middle / last level: a workgroup belongs to ONE bucket of the level above -- composite index o = blockIdx.x / bpb,
// input bucket o % NB_IN of the slices written by the workgroups that served o / NB_IN (`slices_per_group` producer
// waves each; 0 = every producer wave holds a slice of every input bucket: the level below the scan) -- streams those
// slices (256 keys per step, 16-byte loads one step ahead) and re-scatters them with the emitter of pass A.
template <class KS, class E_OUT, uint32_t NB_IN>
__global__ __launch_bounds__(256) void count_rescatter_kernel(const uint32_t* __restrict__ keys_in, uint32_t cap_in,
                                                              const uint32_t* __restrict__ fill_in, uint32_t n_slices_in,
                                                              uint32_t slices_per_group, uint32_t bpb, E_OUT* __restrict__ keys_out,
                                                              uint32_t cap_out, uint32_t* __restrict__ fill_out,
                                                              uint32_t* __restrict__ hist) {
  typedef ScatterEmit<KS, KS::NBITS, 4, E_OUT> Emit;
  Emit::init_lds();
  __syncthreads();
  const uint32_t o = blockIdx.x / bpb, j = blockIdx.x % bpb;
  const uint32_t b_in = slices_per_group ? o % NB_IN : o;
  const uint32_t slice0 = slices_per_group ? (o / NB_IN) * slices_per_group : 0u;
  const uint32_t n_in = slices_per_group ? slices_per_group : n_slices_in;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
  Emit e{keys_out, cap_out, fill_out, hist, wave, lane, blockIdx.x * 4u + wave, o};
  e.bind();
  const uint32_t per = (n_in + bpb - 1) / bpb;
  const uint32_t first = j * per, last = min(n_in, first + per);
  for (uint32_t t = first + wave; t < last; t += 4) {
    const uint32_t sl = slice0 + t;
    const uint32_t n = fill_in[(size_t)sl * NB_IN + b_in];  // multiple of GROUP
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 pad = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    const u4* src4 = reinterpret_cast<const u4*>(keys_in + ((size_t)sl * NB_IN + b_in) * cap_in);
    const uint32_t n4 = n / 4u;
    u4 nxt = lane < n4 ? __builtin_nontemporal_load(&src4[lane]) : pad;
    for (uint32_t i = 0; i < n4; i += 64) {  // wave-uniform trip count
      const u4 k = nxt;
      nxt = (i + 64u + lane < n4) ? __builtin_nontemporal_load(&src4[i + 64u + lane]) : pad;
      e.full(k.x);
      e.full(k.y);
      e.full(k.z);
      e.full(k.w);
    }
  }
  e.drain();
}

// W = 14: table[id] += temp[(bucket1 * 16 + bucket2) * 16 + bucket3][payload3]
__global__ __launch_bounds__(256) void count_gather14_kernel(const uint32_t* __restrict__ temp, uint32_t* __restrict__ hist) {
  const uint32_t np = 1u << 28;
  for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < np; x += gridDim.x * blockDim.x) {
    const uint32_t p1 = Split14L1::payload(x), p2 = Split14L2::payload(p1);
    const uint32_t fb = (Split14L1::bucket(x) * Split14L2::NB + Split14L2::bucket(p1)) * Split14L3::NB + Split14L3::bucket(p2);
    const uint32_t v = temp[((size_t)fb << PAYLOAD_BITS) | Split14L3::payload(p2)];
    if (v) hist[x] += v;
  }
}

// One workgroup = 16 waves = part of one bucket: waves walk the (bucket, producer-wave) slices of their share.
__global__ __launch_bounds__(1024) void count_hist_kernel(const uint16_t* __restrict__ keys, uint32_t slice_cap,
                                                          uint32_t n_slices, uint32_t nb,
                                                          const uint32_t* __restrict__ slice_fill, uint32_t bpb,
                                                          uint32_t* __restrict__ temp, uint32_t slices_per_outer) {
  extern __shared__ uint32_t h[];  // 2^15 bins
  // one level: fine bucket f = b, its slices are all producer waves.  Two levels (slices_per_outer > 0):
  // f = b1 * nb + b2, its slices are the producer waves of the level-2 workgroups that served b1.
  const uint32_t f = blockIdx.x / bpb, j = blockIdx.x % bpb;
  const uint32_t b = slices_per_outer ? f % nb : f;
  const uint32_t slice0 = slices_per_outer ? (f / nb) * slices_per_outer : 0u;
  if (slices_per_outer) n_slices = slices_per_outer;
  for (uint32_t i = threadIdx.x; i < (1u << PAYLOAD_BITS); i += blockDim.x) h[i] = 0;
  __syncthreads();
  const uint32_t per = (n_slices + bpb - 1) / bpb;
  const uint32_t first = slice0 + j * per, last = slice0 + min(n_slices, j * per + per);
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  // (the next slice's fill is asked for while this one's keys are counted: on a small shard -- a rank's eighth of the bench
  // input: ~240 groups per slice -- a slice is two dependent round trips, fill then keys, and the kernel their sum)
  uint32_t fill_next = first + wave < last ? slice_fill[(size_t)(first + wave) * nb + b] : 0u;
  for (uint32_t s = first + wave; s < last; s += 16) {
    const uint32_t n8 = fill_next >> 3;  // groups of 8 keys (16 B); fill is a multiple of 64
    if (s + 16 < last) fill_next = slice_fill[(size_t)(s + 16) * nb + b];
    const uint4* src = reinterpret_cast<const uint4*>(keys + ((size_t)s * nb + b) * slice_cap);
    auto count8 = [&](const uint4& v) {
      const uint32_t k[8] = {v.x & 0xFFFFu, v.x >> 16, v.y & 0xFFFFu, v.y >> 16, v.z & 0xFFFFu, v.z >> 16, v.w & 0xFFFFu, v.w >> 16};
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (k[q] < (1u << PAYLOAD_BITS)) atomicAdd(&h[k[q]], 1u);
    };
    uint32_t i = lane;
    for (; i + 448 < n8; i += 512) {  // eight 16-byte loads in flight per lane before their 64 LDS adds
      uint4 v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {  // streamed once: non-temporal
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(&src[i + 64 * q]));
        v[q] = make_uint4(t.x, t.y, t.z, t.w);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) count8(v[q]);
    }
    if (i < n8) {  // the rest (a whole small slice): up to seven loads, all in flight together (one after the other they were its time)
      uint4 v[7];
#pragma unroll
      for (int q = 0; q < 7; ++q) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 ones = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};  // (keys that count8 skips)
        const u32x4 t = i + 64u * q < n8 ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(&src[i + 64 * q])) : ones;
        v[q] = make_uint4(t.x, t.y, t.z, t.w);
      }
#pragma unroll
      for (int q = 0; q < 7; ++q) count8(v[q]);
    }
  }
  __syncthreads();
  uint32_t* dst = temp + ((size_t)f << PAYLOAD_BITS);
  for (uint32_t i = threadIdx.x; i < (1u << PAYLOAD_BITS); i += blockDim.x) {
    const uint32_t v = h[i];
    if (v) __hip_atomic_fetch_add(&dst[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// W = 12: table[id] += temp[bucket1 * 16 + bucket2][payload2]
__global__ __launch_bounds__(256) void count_gather12_kernel(const uint32_t* __restrict__ temp, uint32_t* __restrict__ hist) {
  const uint32_t np = 1u << 24;
  for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < np; x += gridDim.x * blockDim.x) {
    const uint32_t p1 = Split12L1::payload(x);
    const uint32_t fb = Split12L1::bucket(x) * Split12L2::NB + Split12L2::bucket(p1);
    const uint32_t v = temp[((size_t)fb << PAYLOAD_BITS) | Split12L2::payload(p1)];
    if (v) hist[x] += v;
  }
}

// table[join(bucket, payload)] += temp[bucket][payload]
template <int W, int NBITS>
__global__ __launch_bounds__(256) void count_gather_kernel(const uint32_t* __restrict__ temp, uint32_t np,
                                                           uint32_t* __restrict__ hist) {
  using KS = KeySplit<W, NBITS>;
  for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < np; x += gridDim.x * blockDim.x) {
    const uint32_t v = temp[((size_t)KS::bucket(x) << PAYLOAD_BITS) | KS::payload(x)];
    if (v) hist[x] += v;
  }
}

// ---------------------------------------------------------------------------------------------
// Exact fallback for deferred items (rare: low-complexity runs longer than one item).  One lane per
// item, plain sequential code: replay from `back` windows in front of the item with an empty ring;
// the state is certified once 2(W-1) consecutive windows were not suppressed (or the replay started
// at the head of the run); otherwise quadruple `back`.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t base_at(const uint32_t* __restrict__ words32, uint64_t g) {
  return (words32[g >> 4] >> (2u * (uint32_t)(g & 15u))) & 3u;
}

__global__ __launch_bounds__(64) void count_fixup_kernel(const uint32_t* __restrict__ words32,
                                                         const uint64_t* __restrict__ items, int W, int both,
                                                         uint32_t* __restrict__ hist,
                                                         const uint32_t* __restrict__ defer) {
  const uint32_t n = defer[0];
  const uint32_t mask = (1u << (2 * W)) - 1u;
  const int top = 2 * (W - 1);
  for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) {
    const uint32_t it = defer[1 + q];
    const uint64_t rec = items[it];
    const uint32_t nw = (uint32_t)((rec >> ITEM_NW_SHIFT) & ITEM_NW_MASK);
    const uint64_t ws = rec & ITEM_WS_MASK;
    uint32_t j = it;
    while ((items[j] >> ITEM_CONT_SHIFT) & 1ull) --j;  // head item of the run (its cont bit is 0)
    const uint64_t head = items[j] & ITEM_WS_MASK;
    uint64_t back = 8ull * (uint64_t)(W - 1);
    for (;;) {
      const uint64_t p0 = (ws - head > back) ? ws - back : head;
      bool certified = (p0 == head);
      uint32_t ring[16];
      for (int i = 0; i < 16; ++i) ring[i] = INVALID_ID;
      uint32_t id = 0, rc = 0;
      for (int b = 0; b < W - 1; ++b) {
        const uint32_t c = base_at(words32, p0 + b);
        id = (id >> 2) | (c << top);
        rc = ((rc << 2) & mask) | (c ^ 3u);
      }
      uint32_t clean = 0;
      bool ok = true;
      for (uint64_t t = p0; t < ws + nw; ++t) {  // t = stream offset of the window's first base
        const uint32_t c = base_at(words32, t + (uint64_t)(W - 1));
        id = (id >> 2) | (c << top);
        rc = ((rc << 2) & mask) | (c ^ 3u);
        const uint32_t can = both ? min(id, rc) : id;
        bool match = false;
        for (int d = 1; d <= W - 1; ++d) match |= (ring[(uint32_t)(t - d) & 15u] == can);
        if (t < ws) {
          clean = match ? 0u : clean + 1u;
          if (clean >= 2u * (uint32_t)(W - 1)) certified = true;
        } else {
          if (!certified) {
            ok = false;
            break;
          }
          if (!match) __hip_atomic_fetch_add(&hist[can], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ring[(uint32_t)t & 15u] = match ? INVALID_ID : can;
      }
      if (ok) break;
      back *= 4ull;
    }
  }
}

// count[x] = count[revcomp(x)] for x > revcomp(x)  (src/base_pattern.cpp:387-392)
__global__ __launch_bounds__(256) void mirror_kernel(uint32_t* __restrict__ hist, int W, uint32_t np) {
  for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < np; x += gridDim.x * blockDim.x) {
    const uint32_t r = revcomp32(x, W);
    if (x > r) hist[x] = hist[r];
  }
}

// ---------------------------------------------------------------------------------------------
// K1b: (k+1)-mer counts for inputs made of whole sequences.  Each item owns the bases
// [ws, ws+nw) (+ the W-1 tail bases if it is the last item of its run); per base one 3-mer bin,
// plus first-base / first-2-mer bins at the head of a run; n1 and n2 follow as marginals:
//   n2[ab] = sum_c n3[cab] + #runs starting with ab,   n1[a] = sum_b n2[ba] + #runs starting with a.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bg_count_kernel(const uint32_t* __restrict__ words32,
                                                       const uint64_t* __restrict__ items, uint32_t n_items, int W,
                                                       unsigned long long* __restrict__ out /* 64 + 16 + 4 raw */) {
  __shared__ uint32_t h3[64][65];  // [bin][lane-of-wave-slot], padded: private column per (wave-slot, lane%..)
  __shared__ uint32_t hfirst[20];
  // 256 threads share 65 columns: column = threadIdx.x & 63 -> 4 waves collide on a column, use atomics
  for (int i = threadIdx.x; i < 64 * 65; i += blockDim.x) (&h3[0][0])[i] = 0;
  if (threadIdx.x < 20) hfirst[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t col = threadIdx.x & 63u;
  for (uint32_t it = blockIdx.x * blockDim.x + threadIdx.x; it < n_items; it += gridDim.x * blockDim.x) {
    const uint64_t rec = items[it];
    const uint32_t nw = (uint32_t)((rec >> ITEM_NW_SHIFT) & ITEM_NW_MASK);
    const uint64_t ws = rec & ITEM_WS_MASK;
    const bool cont = ((rec >> ITEM_CONT_SHIFT) & 1ull) != 0;
    const bool last = (it + 1 == n_items) || (((items[it + 1] >> ITEM_CONT_SHIFT) & 1ull) == 0);
    const uint32_t nb = nw + (last ? (uint32_t)(W - 1) : 0u);
    uint32_t y = 0;
    uint32_t pos = 2;  // bases of context available in front of the current one (saturates at 2)
    if (cont) {
      y = (base_at(words32, ws - 2) << 2) | base_at(words32, ws - 1);
    } else {
      pos = 0;
    }
    for (uint32_t b = 0; b < nb; ++b) {
      const uint32_t c = base_at(words32, ws + b);
      y = ((y << 2) | c) & 63u;
      if (pos >= 2) {
        atomicAdd(&h3[y][col], 1u);
      } else if (pos == 1) {
        atomicAdd(&hfirst[4 + (y & 15u)], 1u);
        pos = 2;
      } else {
        atomicAdd(&hfirst[c], 1u);
        pos = 1;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    unsigned long long s = 0;
    for (int i = 0; i < 64; ++i) s += h3[threadIdx.x][i];
    if (s) atomicAdd(&out[threadIdx.x], s);
  } else if (threadIdx.x < 84) {
    const uint32_t v = hfirst[threadIdx.x - 64];
    if (v) atomicAdd(&out[threadIdx.x], (unsigned long long)v);
  }
}

// raw (n3 | first-2-mer | first-base) -> (n1 | n2 | n3) layout of the C ABI
__global__ void bg_finish_kernel(const unsigned long long* __restrict__ raw, unsigned long long* __restrict__ out) {
  const int t = threadIdx.x;
  __shared__ unsigned long long n2[16];
  if (t < 16) {
    unsigned long long s = raw[64 + 4 + t];
    for (int c = 0; c < 4; ++c) s += raw[c * 16 + t];
    n2[t] = s;
    out[4 + t] = s;
  }
  __syncthreads();
  if (t < 4) {
    unsigned long long s = raw[64 + t];
    for (int b = 0; b < 4; ++b) s += n2[b * 4 + t];
    out[t] = s;
  }
  if (t < 64) out[20 + t] = raw[t];
}

// fused K1b: block partials (little-endian 3-mer digits) -> (n1 | n2 | n3) in BaMM order
__global__ void bg_finish_fused_kernel(const uint32_t* __restrict__ partials, uint32_t n_blocks,
                                       unsigned long long* __restrict__ out) {
  __shared__ unsigned long long raw[84];
  __shared__ unsigned long long n3[64];
  __shared__ unsigned long long n2[16];
  __shared__ unsigned long long part[12][84];
  const int t = threadIdx.x;
  {  // 12 strands of 84 threads each sum every 12th block; fixed order -> deterministic
    const int bin = t % 84, strand = t / 84;
    if (strand < 12) {
      // 32 loads in flight per thread (integer sums: any order gives the same value).  The partials were written by
      // workgroups all over the chip and come from HBM: ~2 us per round trip, and one dependent load after the other made
      // this one-workgroup kernel 25 us long.
      unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      uint32_t b = strand;
      for (; b + 31u * 12u < n_blocks; b += 32u * 12u) {
        uint32_t v[32];
#pragma unroll
        for (uint32_t j = 0; j < 32u; ++j) v[j] = partials[(size_t)(b + 12u * j) * 84 + bin];
#pragma unroll
        for (uint32_t j = 0; j < 32u; ++j) acc[j & 7u] += v[j];
      }
      for (; b < n_blocks; b += 12) acc[0] += partials[(size_t)b * 84 + bin];
      part[strand][bin] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    }
  }
  __syncthreads();
  if (t < 84) {
    unsigned long long s = 0;
    for (int k = 0; k < 12; ++k) s += part[k][t];
    raw[t] = s;
  }
  __syncthreads();
  if (t < 64) {  // t = BaMM index d0*16 + d1*4 + d2 (d0 oldest)
    const int d0 = t >> 4, d1 = (t >> 2) & 3, d2 = t & 3;
    n3[t] = raw[d0 | (d1 << 2) | (d2 << 4)];
    out[20 + t] = n3[t];
  }
  __syncthreads();
  if (t < 16) {  // t = a*4 + b
    const int a = t >> 2, b = t & 3;
    unsigned long long s = raw[68 + (a | (b << 2))];
    for (int c = 0; c < 4; ++c) s += n3[c * 16 + t];
    n2[t] = s;
    out[4 + t] = s;
  }
  __syncthreads();
  if (t < 4) {
    unsigned long long s = raw[64 + t];
    for (int b = 0; b < 4; ++b) s += n2[b * 4 + t];
    out[t] = s;
  }
}

// BackgroundModel::calculateV (src/shared/BackgroundModel.cpp:490-530), one thread, float32 in the
// reference's order.  Counts are converted integer -> float with round-to-nearest (the reference's
// `int` counters overflow beyond 2^31 bases; below that the results are bit-identical).
__global__ __launch_bounds__(128) void bg_model_kernel(const unsigned long long* __restrict__ n, int K, float a0, float a1, float a2,
                                                       float* __restrict__ V) {
  // one workgroup; every entry and every group of four is computed by its own thread with the reference's operations
  // in the reference's order (a single thread walking all 84 entries took 21 us of dependent loads and divisions)
  __shared__ float cnt[84];
  __shared__ float v[84];
  const int t = threadIdx.x;
  const float alpha[3] = {a0, a1, a2};
  const int off[3] = {0, 4, 20};
  if (t < 84) cnt[t] = (float)n[t];
  __shared__ float base_f;
  if (t == 0) {
    unsigned long long base_counts = 0;
    for (int y = 0; y < 4; ++y) base_counts += n[y];
    base_f = (float)base_counts;
  }
  __syncthreads();
  if (t < 4) v[t] = (cnt[t] + alpha[0] * 0.25f) / (base_f + alpha[0]);
  __syncthreads();
  for (int k = 1; k <= K; ++k) {
    const int ny = 1 << (2 * (k + 1));
    const int yk = 1 << (2 * k);
    if (t < ny) v[off[k] + t] = (cnt[off[k] + t] + alpha[k] * v[off[k - 1] + t % yk]) / (cnt[off[k - 1] + t / 4] + alpha[k]);
    __syncthreads();
    if (t < ny / 4) {
      float* g = v + off[k] + 4 * t;
      float factor = 0.0f;
      for (int a = 0; a < 4; ++a) factor += g[a];
      for (int a = 0; a < 4; ++a) g[a] /= factor;
    }
    __syncthreads();
  }
  const int used = off[K] + (1 << (2 * (K + 1)));
  if (t < 84) V[t] = t < used ? v[t] : 0.0f;
}

// ---------------------------------------------------------------------------------------------
// Synthetic input written directly in the packed layout.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

__global__ __launch_bounds__(256) void synth_words_kernel(uint64_t seed, uint64_t seq0, uint64_t n_seq, uint32_t L,
                                                          uint64_t* __restrict__ words, uint64_t n_words) {
  const uint64_t motif = 2ull | (1ull << 2) | (3ull << 4) | (2ull << 6) | (0ull << 8) | (2ull << 10) | (3ull << 12) |
                         (1ull << 14) | (0ull << 16) | (3ull << 18);  // GCTGAGTCAT
  const uint64_t total = n_seq * (uint64_t)L;
  for (uint64_t w = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t v = 0;
    for (int k = 0; k < 32; ++k) {
      const uint64_t g = w * 32 + k;
      if (g < PENGK_FRONT_PAD_BASES) continue;
      const uint64_t li = g - PENGK_FRONT_PAD_BASES;
      if (li >= total) break;
      const uint64_t n = seq0 + li / L;
      const uint32_t j = (uint32_t)(li % L);
      uint64_t d = mix64(seed + 0x9E3779B97F4A7C15ull * (n * (uint64_t)L + j + 1)) >> 62;
      if (L >= 10 && mix64(seed ^ 0xA5A5A5A5ull ^ (n + 1)) % 10 == 0) {
        const uint64_t q = mix64(seed ^ 0x5A5A5A5Aull ^ (n + 1)) % (L - 9);
        if (j >= q && j < q + 10) d = (motif >> (2 * (j - q))) & 3ull;
      }
      v |= d << (2 * k);
    }
    words[w] = v;
  }
}

__global__ __launch_bounds__(256) void synth_items_kernel(uint64_t n_seq, uint32_t L, int W, uint32_t M,
                                                          uint64_t* __restrict__ items) {
  const uint64_t nwin = (uint64_t)L - W + 1;
  const uint64_t per = (nwin + M - 1) / M;
  const uint64_t n = n_seq * per;
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t s = i / per;
    const uint64_t f = (i % per) * M;
    const uint64_t nw = nwin - f < M ? nwin - f : M;
    items[i] = (PENGK_FRONT_PAD_BASES + s * L + f) | (nw << ITEM_NW_SHIFT) | ((uint64_t)(f ? 1 : 0) << ITEM_CONT_SHIFT);
  }
}

// kernel launch with the (BOTH, BG) template bits chosen at run time
#define PENGK_LAUNCH_BB(KERNEL, TARGS, both, bg, grid, block, ...)                                              \
  do {                                                                                                         \
    if (both) {                                                                                                \
      if (bg) hipLaunchKernelGGL((KERNEL<TARGS(true, true)>), grid, block, 0, ctx->stream, __VA_ARGS__);        \
      else hipLaunchKernelGGL((KERNEL<TARGS(true, false)>), grid, block, 0, ctx->stream, __VA_ARGS__);          \
    } else {                                                                                                   \
      if (bg) hipLaunchKernelGGL((KERNEL<TARGS(false, true)>), grid, block, 0, ctx->stream, __VA_ARGS__);       \
      else hipLaunchKernelGGL((KERNEL<TARGS(false, false)>), grid, block, 0, ctx->stream, __VA_ARGS__);         \
    }                                                                                                          \
  } while (0)

// block partials of the fused K1b (84 uint32 per block of the scan grid)
int bg_partials_buffer(pengk_ctx* ctx, uint32_t blocks, uint32_t** out) {
  int rc = ensure_scratch(ctx, &ctx->d_bg_partials, &ctx->bg_partials_bytes, (size_t)blocks * 84 * sizeof(uint32_t));
  *out = (uint32_t*)ctx->d_bg_partials;
  return rc;
}

int bg_finish_fused(pengk_ctx* ctx, uint32_t blocks, uint64_t* d_bg) {
  hipLaunchKernelGGL(bg_finish_fused_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t*)ctx->d_bg_partials, blocks,
                     (unsigned long long*)d_bg);
  PENGK_HIP(hipGetLastError());
  return PENGK_OK;
}

template <int W>
int launch_direct_w(pengk_ctx* ctx, int both, uint32_t* d_counts, uint64_t* d_ltot, uint32_t n_items, uint64_t* d_bg) {
  const uint32_t* w32 = (const uint32_t*)ctx->d_words;
  const uint32_t blocks_needed = (n_items + 255) / 256;
  const uint32_t max_blocks = (uint32_t)ctx->num_cu * 8u;
  const uint32_t blocks = blocks_needed < max_blocks ? blocks_needed : max_blocks;
  unsigned long long* lt = (unsigned long long*)d_ltot;
  uint32_t* bgp = nullptr;
  if (d_bg) {
    int rc = bg_partials_buffer(ctx, blocks, &bgp);
    if (rc) return rc;
  }
#define TA_DIRECT(B, G) W, B, G
  PENGK_LAUNCH_BB(count_kernel, TA_DIRECT, both, d_bg != nullptr, dim3(blocks), dim3(256), w32, ctx->d_items, n_items, d_counts,
                  lt, ctx->d_defer, bgp);
#undef TA_DIRECT
  PENGK_HIP(hipGetLastError());
  return d_bg ? bg_finish_fused(ctx, blocks, d_bg) : PENGK_OK;
}

template <int W>
int launch_partition_w(pengk_ctx* ctx, int both, uint32_t* d_counts, uint64_t* d_ltot, uint32_t n_items, uint64_t* d_bg) {
  constexpr int NBITS = 2 * W - PAYLOAD_BITS;
  constexpr uint32_t NB = 1u << NBITS;
  const uint32_t np = 1u << (2 * W);
  const uint32_t* w32 = (const uint32_t*)ctx->d_words;
  // grid of pass A: as many workgroups per CU as the LDS admits (8.3 KiB of rings per wave at NB = 32)
  constexpr uint32_t TPB = 64u * SCATTER_WPW;
  const uint32_t blocks_needed = (n_items + TPB - 1) / TPB;
  constexpr uint32_t lds_per_wg = (uint32_t)(sizeof(ScatterShared<NBITS, SCATTER_WPW>) + sizeof(BgLds) + 511u) & ~511u;
  const uint32_t per_cu = ctx->scatter_blocks_per_cu ? (uint32_t)ctx->scatter_blocks_per_cu : (160u * 1024u) / lds_per_wg;
  const uint32_t max_blocks = (uint32_t)ctx->num_cu * per_cu;
  const uint32_t blocks = blocks_needed < max_blocks ? blocks_needed : max_blocks;
  const uint32_t n_waves = blocks * (uint32_t)SCATTER_WPW;
  // static slices region[wave][bucket]: expected share + 50 % + slack, in groups of 64 entries
  uint64_t windows = ctx->n_windows_hint ? ctx->n_windows_hint : ctx->n_items * (uint64_t)ctx->item_windows;
  const uint64_t share = windows / ((uint64_t)NB * n_waves);
  uint64_t cap64 = share + share / 2 + 512;  // 1.5x the uniform share: real genomes are not uniform
  if (ctx->key_cap_override) cap64 = ctx->key_cap_override;  // test hook: force slices to overflow
  cap64 = (cap64 + 63) / 64 * 64;
  if (cap64 * NB >= (1ull << 31))  // byte offsets inside a wave's NB slices are 32-bit
    return fail(PENGK_ERR_RANGE, "shard too large for the partitioned count (slice of %llu keys)", (unsigned long long)cap64);
  const uint32_t slice_cap = (uint32_t)cap64;
  int rc = ensure_scratch(ctx, &ctx->d_keys, &ctx->keys_bytes, (size_t)NB * n_waves * slice_cap * sizeof(uint16_t));
  if (rc) return rc;
  const size_t fill_words = ((size_t)NB * n_waves + 63) / 64 * 64;
  const size_t aux_need = fill_words * sizeof(uint32_t) + (size_t)np * sizeof(uint32_t);  // slice fills | bucket-major table
  rc = ensure_scratch(ctx, &ctx->d_count_aux, &ctx->count_aux_bytes, aux_need);
  if (rc) return rc;
  uint32_t* slice_fill = (uint32_t*)ctx->d_count_aux;
  uint32_t* temp = slice_fill + fill_words;
  PENGK_HIP(hipMemsetAsync(ctx->d_count_aux, 0, aux_need, ctx->stream));
  unsigned long long* lt = (unsigned long long*)d_ltot;
  uint16_t* keys = (uint16_t*)ctx->d_keys;
  uint32_t* bgp = nullptr;
  if (d_bg) {
    rc = bg_partials_buffer(ctx, blocks, &bgp);
    if (rc) return rc;
  }
#define TA_SCATTER(B, G) W, B, NBITS, G
  PENGK_LAUNCH_BB(count_scatter_kernel, TA_SCATTER, both, d_bg != nullptr, dim3(blocks), dim3(TPB), w32, ctx->d_items, n_items,
                  keys, slice_cap, slice_fill, d_counts, lt, ctx->d_defer, bgp);
#undef TA_SCATTER
  PENGK_HIP(hipGetLastError());
  if (d_bg) {
    rc = bg_finish_fused(ctx, blocks, d_bg);
    if (rc) return rc;
  }
  // pass B: one 1024-thread workgroup per CU (128 KiB of LDS), bpb workgroups per bucket
  uint32_t bpb = ((uint32_t)ctx->num_cu * 2u + NB - 1) / NB;
  const uint64_t per_bucket = windows / NB + 1;
  const uint32_t useful = (uint32_t)((per_bucket + 65535) / 65536);  // >= 64 Ki keys per workgroup or it is not worth a block
  if (bpb > useful) bpb = useful;
  if (bpb > n_waves) bpb = n_waves;
  if (bpb < 1) bpb = 1;
  hipLaunchKernelGGL(count_hist_kernel, dim3(NB * bpb), dim3(1024), 4 << PAYLOAD_BITS, ctx->stream, keys, slice_cap, n_waves,
                     NB, slice_fill, bpb, temp, 0u);
  PENGK_HIP(hipGetLastError());
  const uint32_t gb = (np + 255) / 256 < 2048u ? (np + 255) / 256 : 2048u;
  hipLaunchKernelGGL((count_gather_kernel<W, NBITS>), dim3(gb), dim3(256), 0, ctx->stream, temp, np, d_counts);
  PENGK_HIP(hipGetLastError());
  return PENGK_OK;
}

// W = 12: scan -> 32 coarse buckets of 32-bit keys -> 16 fine buckets each of 16-bit keys -> 512 LDS histograms
int launch_partition12(pengk_ctx* ctx, int both, uint32_t* d_counts, uint64_t* d_ltot, uint32_t n_items, uint64_t* d_bg) {
  const uint32_t np = 1u << 24;
  const uint32_t* w32 = (const uint32_t*)ctx->d_words;
  constexpr uint32_t TPB1 = 64u * SCATTER12L1_WPW;
  constexpr uint32_t NB1 = Split12L1::NB, NB2 = Split12L2::NB;
  const uint32_t blocks_needed = (n_items + TPB1 - 1) / TPB1;
  constexpr uint32_t lds_per_wg1 = (uint32_t)(sizeof(ScatterShared<Split12L1::NBITS, SCATTER12L1_WPW, uint32_t>) + sizeof(BgLds) + 511u) & ~511u;
  const uint32_t max_blocks = (uint32_t)ctx->num_cu * ((160u * 1024u) / lds_per_wg1);
  const uint32_t blocks1 = blocks_needed < max_blocks ? blocks_needed : max_blocks;
  const uint32_t n_waves1 = blocks1 * (uint32_t)SCATTER12L1_WPW;
  const uint64_t windows = ctx->n_windows_hint ? ctx->n_windows_hint : ctx->n_items * (uint64_t)ctx->item_windows;
  // level-1 slices
  uint64_t share1 = windows / ((uint64_t)NB1 * n_waves1);
  uint64_t cap1_64 = share1 + share1 / 2 + 256;
  if (ctx->key_cap_override) cap1_64 = ctx->key_cap_override;
  cap1_64 = (cap1_64 + 63) / 64 * 64;
  // level-2 grid: bpb1 workgroups per level-1 bucket
  uint32_t bpb1 = ((uint32_t)ctx->num_cu * 8u + NB1 - 1u) / NB1;
  if (bpb1 > n_waves1) bpb1 = n_waves1;
  if (bpb1 < 1) bpb1 = 1;
  const uint32_t blocks2 = NB1 * bpb1;
  const uint32_t n_waves2 = blocks2 * 4u;
  uint64_t share2 = windows / ((uint64_t)NB2 * n_waves2);
  uint64_t cap2_64 = share2 + share2 / 2 + 512;
  if (ctx->key_cap_override) cap2_64 = ctx->key_cap_override;
  cap2_64 = (cap2_64 + 63) / 64 * 64;
  if (cap1_64 * NB1 >= (1ull << 30) || cap2_64 * NB2 >= (1ull << 31)) return fail(  // 32-bit byte offsets inside a wave's slices
      PENGK_ERR_RANGE, "shard too large for the partitioned count");
  const uint32_t cap1 = (uint32_t)cap1_64, cap2 = (uint32_t)cap2_64;
  const size_t bytes1 = (size_t)n_waves1 * NB1 * cap1 * sizeof(uint32_t);
  const size_t bytes2 = (size_t)n_waves2 * NB2 * cap2 * sizeof(uint16_t);
  int rc = ensure_scratch(ctx, &ctx->d_keys, &ctx->keys_bytes, bytes1 + bytes2);
  if (rc) return rc;
  uint32_t* keys1 = (uint32_t*)ctx->d_keys;
  uint16_t* keys2 = (uint16_t*)((char*)ctx->d_keys + bytes1);
  const size_t fill1_words = ((size_t)n_waves1 * NB1 + 63) / 64 * 64, fill2_words = ((size_t)n_waves2 * NB2 + 63) / 64 * 64;
  const size_t aux_need = (fill1_words + fill2_words + (size_t)np) * sizeof(uint32_t);
  rc = ensure_scratch(ctx, &ctx->d_count_aux, &ctx->count_aux_bytes, aux_need);
  if (rc) return rc;
  uint32_t* fill1 = (uint32_t*)ctx->d_count_aux;
  uint32_t* fill2 = fill1 + fill1_words;
  uint32_t* temp = fill2 + fill2_words;
  PENGK_HIP(hipMemsetAsync(ctx->d_count_aux, 0, aux_need, ctx->stream));
  unsigned long long* lt = (unsigned long long*)d_ltot;
  uint32_t* bgp = nullptr;
  if (d_bg) {
    rc = bg_partials_buffer(ctx, blocks1, &bgp);
    if (rc) return rc;
  }
#define TA_S12(B, G) B, G
  PENGK_LAUNCH_BB(count_scatter12_kernel, TA_S12, both, d_bg != nullptr, dim3(blocks1), dim3(TPB1), w32, ctx->d_items, n_items, keys1,
                  cap1, fill1, d_counts, lt, ctx->d_defer, bgp);
#undef TA_S12
  PENGK_HIP(hipGetLastError());
  if (d_bg) {
    rc = bg_finish_fused(ctx, blocks1, d_bg);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(count_rescatter12_kernel, dim3(blocks2), dim3(256), 0, ctx->stream, keys1, cap1, fill1, n_waves1, bpb1, keys2,
                     cap2, fill2, d_counts);
  PENGK_HIP(hipGetLastError());
  hipLaunchKernelGGL(count_hist_kernel, dim3(512), dim3(1024), 4 << PAYLOAD_BITS, ctx->stream, keys2, cap2, n_waves2, NB2, fill2, 1u,
                     temp, bpb1 * 4u);
  PENGK_HIP(hipGetLastError());
  hipLaunchKernelGGL(count_gather12_kernel, dim3(4096), dim3(256), 0, ctx->stream, temp, d_counts);
  PENGK_HIP(hipGetLastError());
  return PENGK_OK;
}

// W = 14: scan -> 16 buckets of 32-bit keys -> 16 each of 32-bit keys -> 32 each of 16-bit keys -> 8192 LDS histograms.
// 20 B per window through HBM (4 + 4 written and read, 2 written and read) instead of one device-scope atomic per window
// on a 1 GiB table.
int launch_partition14(pengk_ctx* ctx, int both, uint32_t* d_counts, uint64_t* d_ltot, uint32_t n_items, uint64_t* d_bg) {
  const size_t np = (size_t)1 << 28;
  const uint32_t* w32 = (const uint32_t*)ctx->d_words;
  constexpr uint32_t TPB1 = 64u * SCATTER12L1_WPW;
  constexpr uint32_t NB1 = Split14L1::NB, NB2 = Split14L2::NB, NB3 = Split14L3::NB;
  static_assert(NB1 * NB2 * NB3 == 8192, "2^13 LDS histograms");
  const uint32_t blocks_needed = (n_items + TPB1 - 1) / TPB1;
  constexpr uint32_t lds_per_wg1 = (uint32_t)(sizeof(ScatterShared<Split14L1::NBITS, SCATTER12L1_WPW, uint32_t>) + sizeof(BgLds) + 511u) & ~511u;
  const uint32_t max_blocks = (uint32_t)ctx->num_cu * ((160u * 1024u) / lds_per_wg1);
  const uint32_t blocks1 = blocks_needed < max_blocks ? blocks_needed : max_blocks;
  const uint32_t n_waves1 = blocks1 * (uint32_t)SCATTER12L1_WPW;
  const uint64_t windows = ctx->n_windows_hint ? ctx->n_windows_hint : ctx->n_items * (uint64_t)ctx->item_windows;
  auto cap_of = [&](uint64_t slices, uint64_t slack) {  // 1.5 x the uniform share of a (wave, bucket) slice, whole groups
    uint64_t c = windows / slices;
    c = c + c / 2 + slack;
    if (ctx->key_cap_override) c = ctx->key_cap_override;
    return (c + 63) / 64 * 64;
  };
  // level 2: bpb1 workgroups per level-1 bucket; level 3: bpb2 workgroups per (level-1, level-2) bucket
  uint32_t bpb1 = ((uint32_t)ctx->num_cu * 8u + NB1 - 1u) / NB1;
  if (bpb1 > n_waves1) bpb1 = n_waves1;
  if (bpb1 < 1) bpb1 = 1;
  const uint32_t blocks2 = NB1 * bpb1, n_waves2 = blocks2 * 4u;
  uint32_t bpb2 = ((uint32_t)ctx->num_cu * 8u + NB1 * NB2 - 1u) / (NB1 * NB2);
  if (bpb2 > bpb1 * 4u) bpb2 = bpb1 * 4u;
  if (bpb2 < 1) bpb2 = 1;
  const uint32_t blocks3 = NB1 * NB2 * bpb2, n_waves3 = blocks3 * 4u;
  const uint64_t cap1_64 = cap_of((uint64_t)NB1 * n_waves1, 256), cap2_64 = cap_of((uint64_t)NB2 * n_waves2, 256), cap3_64 = cap_of((uint64_t)NB3 * n_waves3, 256);
  if (cap1_64 * NB1 >= (1ull << 30) || cap2_64 * NB2 >= (1ull << 30) || cap3_64 * NB3 >= (1ull << 31))  // 32-bit byte offsets
    return fail(PENGK_ERR_RANGE, "shard too large for the partitioned count");
  const uint32_t cap1 = (uint32_t)cap1_64, cap2 = (uint32_t)cap2_64, cap3 = (uint32_t)cap3_64;
  const size_t bytes1 = (size_t)n_waves1 * NB1 * cap1 * sizeof(uint32_t);
  const size_t bytes2 = (size_t)n_waves2 * NB2 * cap2 * sizeof(uint32_t);
  const size_t bytes3 = (size_t)n_waves3 * NB3 * cap3 * sizeof(uint16_t);
  int rc = ensure_scratch(ctx, &ctx->d_keys, &ctx->keys_bytes, bytes1 + bytes2 + bytes3);
  if (rc) return rc;
  uint32_t* keys1 = (uint32_t*)ctx->d_keys;
  uint32_t* keys2 = (uint32_t*)((char*)ctx->d_keys + bytes1);
  uint16_t* keys3 = (uint16_t*)((char*)ctx->d_keys + bytes1 + bytes2);
  const size_t f1 = ((size_t)n_waves1 * NB1 + 63) / 64 * 64, f2 = ((size_t)n_waves2 * NB2 + 63) / 64 * 64,
               f3 = ((size_t)n_waves3 * NB3 + 63) / 64 * 64;
  const size_t aux_need = (f1 + f2 + f3 + np) * sizeof(uint32_t);
  rc = ensure_scratch(ctx, &ctx->d_count_aux, &ctx->count_aux_bytes, aux_need);
  if (rc) return rc;
  uint32_t* fill1 = (uint32_t*)ctx->d_count_aux;
  uint32_t* fill2 = fill1 + f1;
  uint32_t* fill3 = fill2 + f2;
  uint32_t* temp = fill3 + f3;
  PENGK_HIP(hipMemsetAsync(ctx->d_count_aux, 0, aux_need, ctx->stream));
  unsigned long long* lt = (unsigned long long*)d_ltot;
  uint32_t* bgp = nullptr;
  if (d_bg) {
    rc = bg_partials_buffer(ctx, blocks1, &bgp);
    if (rc) return rc;
  }
#define TA_S14(B, G) B, G
  PENGK_LAUNCH_BB(count_scatter14_kernel, TA_S14, both, d_bg != nullptr, dim3(blocks1), dim3(TPB1), w32, ctx->d_items, n_items, keys1,
                  cap1, fill1, d_counts, lt, ctx->d_defer, bgp);
#undef TA_S14
  PENGK_HIP(hipGetLastError());
  if (d_bg) {
    rc = bg_finish_fused(ctx, blocks1, d_bg);
    if (rc) return rc;
  }
  hipLaunchKernelGGL((count_rescatter_kernel<Split14L2, uint32_t, NB1>), dim3(blocks2), dim3(256), 0, ctx->stream, keys1, cap1, fill1,
                     n_waves1, 0u, bpb1, keys2, cap2, fill2, d_counts);
  PENGK_HIP(hipGetLastError());
  hipLaunchKernelGGL((count_rescatter_kernel<Split14L3, uint16_t, NB2>), dim3(blocks3), dim3(256), 0, ctx->stream, keys2, cap2, fill2,
                     n_waves2, bpb1 * 4u, bpb2, keys3, cap3, fill3, d_counts);
  PENGK_HIP(hipGetLastError());
  hipLaunchKernelGGL(count_hist_kernel, dim3(8192), dim3(1024), 4 << PAYLOAD_BITS, ctx->stream, keys3, cap3, n_waves3, NB3, fill3, 1u,
                     temp, bpb2 * 4u);
  PENGK_HIP(hipGetLastError());
  hipLaunchKernelGGL(count_gather14_kernel, dim3(8192), dim3(256), 0, ctx->stream, temp, d_counts);
  PENGK_HIP(hipGetLastError());
  return PENGK_OK;
}

template <int W>
int launch_count_w(pengk_ctx* ctx, int both, uint32_t* d_counts, uint64_t* d_ltot, uint64_t* d_bg) {
  const uint32_t n_items = (uint32_t)ctx->n_items;
  if (n_items == 0) {
    if (d_bg) PENGK_HIP(hipMemsetAsync(d_bg, 0, 84 * sizeof(uint64_t), ctx->stream));
    return PENGK_OK;
  }
  if constexpr (W < 4) {
    // the fused K1b reads the 3-mers off the rolling id's top three digits: an id of two digits has none -- the
    // stand-alone background count runs beside the scan
    if (d_bg) {
      const int rc_bg = launch_bg_count(ctx, d_bg);
      if (rc_bg) return rc_bg;
      d_bg = nullptr;
    }
  }
  int impl = ctx->count_impl;
  constexpr bool can_partition = (W == 8 || W == 10 || W == 12 || W == 14);
  if (impl == 0) impl = can_partition ? 2 : 1;
  if (impl == 2 && !can_partition) return fail(PENGK_ERR_UNSUPPORTED, "partitioned count is built for W = 8 .. 14 only");
  int rc;
  if constexpr (W == 14) {
    rc = impl == 2 ? launch_partition14(ctx, both, d_counts, d_ltot, n_items, d_bg)
                   : launch_direct_w<W>(ctx, both, d_counts, d_ltot, n_items, d_bg);
  } else if constexpr (W == 12) {
    rc = impl == 2 ? launch_partition12(ctx, both, d_counts, d_ltot, n_items, d_bg)
                   : launch_direct_w<W>(ctx, both, d_counts, d_ltot, n_items, d_bg);
  } else if constexpr (can_partition) {
    rc = impl == 2 ? launch_partition_w<W>(ctx, both, d_counts, d_ltot, n_items, d_bg)
                   : launch_direct_w<W>(ctx, both, d_counts, d_ltot, n_items, d_bg);
  } else {
    rc = launch_direct_w<W>(ctx, both, d_counts, d_ltot, n_items, d_bg);
  }
  if (rc) return rc;
  hipLaunchKernelGGL(count_fixup_kernel, dim3(64), dim3(64), 0, ctx->stream, (const uint32_t*)ctx->d_words, ctx->d_items, W,
                     both, d_counts, ctx->d_defer);
  PENGK_HIP(hipGetLastError());
  return PENGK_OK;
}

}  // namespace

// hipFuncSetAttribute applies to the CURRENT device: called from pengk_create after hipSetDevice, once per context
int count_init_device() {
  PENGK_HIP(hipFuncSetAttribute((const void*)count_hist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 << PAYLOAD_BITS));
  return PENGK_OK;
}

int launch_count(pengk_ctx* ctx, int both, uint32_t* d_counts, uint64_t* d_ltot, uint64_t* d_bg) {
  const int W = ctx->W;
  const size_t np = (size_t)1 << (2 * W);
  // defer list: counter + one slot per item
  {
    const size_t need = (ctx->n_items + 2) * sizeof(uint32_t);
    size_t have = ctx->defer_cap * sizeof(uint32_t);
    if (have < need) {
      int rc = ensure_scratch(ctx, (void**)&ctx->d_defer, &have, need);
      if (rc) return rc;
      ctx->defer_cap = have / sizeof(uint32_t);
    }
  }
  PENGK_HIP(hipMemsetAsync(ctx->d_defer, 0, sizeof(uint32_t), ctx->stream));
  PENGK_HIP(hipMemsetAsync(d_counts, 0, np * sizeof(uint32_t), ctx->stream));
  PENGK_HIP(hipMemsetAsync(d_ltot, 0, sizeof(uint64_t), ctx->stream));
  switch (W) {
    case 2: return launch_count_w<2>(ctx, both, d_counts, d_ltot, d_bg);
    case 4: return launch_count_w<4>(ctx, both, d_counts, d_ltot, d_bg);
    case 6: return launch_count_w<6>(ctx, both, d_counts, d_ltot, d_bg);
    case 8: return launch_count_w<8>(ctx, both, d_counts, d_ltot, d_bg);
    case 10: return launch_count_w<10>(ctx, both, d_counts, d_ltot, d_bg);
    case 12: return launch_count_w<12>(ctx, both, d_counts, d_ltot, d_bg);
    case 14: return launch_count_w<14>(ctx, both, d_counts, d_ltot, d_bg);
    default: return fail(PENGK_ERR_ARG, "pattern length %d unsupported", W);
  }
}

int launch_mirror(pengk_ctx* ctx, int W, uint32_t* d_counts) {
  const uint32_t np = 1u << (2 * W);
  const uint32_t blocks = (np + 255) / 256 < 4096u ? (np + 255) / 256 : 4096u;
  hipLaunchKernelGGL(mirror_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_counts, W, np);
  PENGK_HIP(hipGetLastError());
  return PENGK_OK;
}

int launch_bg_count(pengk_ctx* ctx, uint64_t* d_bg) {
  int rc = ensure_scratch(ctx, &ctx->d_misc, &ctx->misc_bytes, 84 * sizeof(uint64_t));
  if (rc) return rc;
  unsigned long long* raw = (unsigned long long*)ctx->d_misc;
  PENGK_HIP(hipMemsetAsync(raw, 0, 84 * sizeof(uint64_t), ctx->stream));
  const uint32_t n_items = (uint32_t)ctx->n_items;
  if (n_items) {
    const uint32_t need = (n_items + 255) / 256;
    const uint32_t blocks = need < (uint32_t)ctx->num_cu * 4u ? need : (uint32_t)ctx->num_cu * 4u;
    hipLaunchKernelGGL(bg_count_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const uint32_t*)ctx->d_words,
                       ctx->d_items, n_items, ctx->W, raw);
    PENGK_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(bg_finish_kernel, dim3(1), dim3(64), 0, ctx->stream, raw, (unsigned long long*)d_bg);
  PENGK_HIP(hipGetLastError());
  return PENGK_OK;
}

int launch_bg_model(pengk_ctx* ctx, const uint64_t* d_bg, int K, const float* h_alpha, float* d_V) {
  hipLaunchKernelGGL(bg_model_kernel, dim3(1), dim3(128), 0, ctx->stream, (const unsigned long long*)d_bg, K, h_alpha[0],
                     h_alpha[1], h_alpha[2], d_V);
  PENGK_HIP(hipGetLastError());
  return PENGK_OK;
}

int launch_synth(pengk_ctx* ctx, uint64_t seed, uint64_t seq0, uint64_t n_seq, uint32_t L, int W, int item_windows,
                 uint64_t* d_words, uint64_t* d_items) {
  uint64_t nw = 0, ni = 0;
  int rc = pengk_synth_sizes(n_seq, L, W, item_windows, &nw, &ni);
  if (rc) return rc;
  hipLaunchKernelGGL(synth_words_kernel, dim3(4096), dim3(256), 0, ctx->stream, seed, seq0, n_seq, L, d_words, nw);
  PENGK_HIP(hipGetLastError());
  if (ni) {
    hipLaunchKernelGGL(synth_items_kernel, dim3(2048), dim3(256), 0, ctx->stream, n_seq, L, W, (uint32_t)item_windows, d_items);
    PENGK_HIP(hipGetLastError());
  }
  return PENGK_OK;
}

}  // namespace pengk
