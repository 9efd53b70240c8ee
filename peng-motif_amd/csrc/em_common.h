// em_common.h -- what the EM's translation units share (em.hip: the product's kernels; em_legacy.hip: the serial mode's
// earlier generations, kept for pattern lengths below 10 and as independent cross-checks in the tests).
#pragma once
#include <algorithm>
#include <type_traits>

#include "pengk_internal.h"
#include "seqsum.h"

namespace pengk {
namespace {

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

// The terms of a cell, for the scans of seqsum.h: term c of cell (p, a) is the weight of the x whose digit p is a, ascending.
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

__global__ void em_init_kernel(int n, int W, float threshold, int max_it, int32_t* __restrict__ state,
                               float* __restrict__ change) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float c0 = (float)W;  // `float change = pattern_length` (src/peng.cpp:101)
  state[2 * i] = 0;
  state[2 * i + 1] = !(c0 <= threshold || 0 >= max_it);
  change[i] = c0;
}

}  // namespace

// em_legacy.hip: the serial mode (em_fast = 2) by the earlier generations -- generation 0: one dependent addition after
// the other (em_fold_kernel), 1: the scan of seqsum.h block after block (em_fold_scan_kernel; cells of at least four
// blocks, i.e. W >= 8).  The product takes them for W <= 8; for longer patterns they are reachable through the test
// hook pengk_test_em_generation only.
int launch_em_serial_legacy(pengk_ctx* ctx, int W, int generation, int64_t n_pwm, float* d_pwms, float saturation, float threshold,
                            int max_it, const uint32_t* d_counts, const float* d_bg, int32_t* d_state, float* d_change, size_t budget);

}  // namespace pengk
