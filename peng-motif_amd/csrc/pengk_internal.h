// pengk_internal.h -- shared declarations of the gfx950 hot-path library (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pengk.h"

struct pengk_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int num_cu = 256;
  // attached sequences (caller-owned device memory)
  const uint64_t* d_words = nullptr;
  uint64_t n_words = 0;
  const uint64_t* d_items = nullptr;
  uint64_t n_items = 0;
  int W = 0;
  int item_windows = 0;
  uint64_t max_bin_bound = 0;
  int all_whole = 0;
  // scratch owned by the context
  uint32_t* d_defer = nullptr;  // deferred item indices + counter in slot 0
  uint64_t defer_cap = 0;
  double* d_em_partials = nullptr;
  size_t em_partials_bytes = 0;
  float* d_em_tables = nullptr;  // K5 fast mode: count*saturation | saturation*background, 4^W floats each
  size_t em_tables_bytes = 0;
  void* d_em_blocks = nullptr;   // K5 serial mode, blocks ahead of their chain: block sums | block records (em.hip)
  size_t em_blocks_bytes = 0;
  void* d_pair_mids = nullptr;   // K2+K3 twin tiles: the middles m <= rc(m), one workgroup each (stats.hip)
  size_t pair_mids_bytes = 0;
  int pair_mids_W = 0;           // the pattern length that list was made for
  void* d_em_look = nullptr;     // K5 serial mode, two launches per iteration: the look-back words of the spans (em.hip)
  size_t em_look_bytes = 0;
  uint32_t em_epoch = 0;         // ... and the epoch of the last launch that wrote them
  int em_test_lookback = 0;      // test hook: every n-th workgroup of em_span_fused_kernel acts as if its look-back had timed out
  unsigned long long* d_em_counters = nullptr;  // K5 serial mode: what the chains of the last pengk_em call met (seqsum::WalkCounts)
  hipStream_t em_streams[3] = {nullptr, nullptr, nullptr};  // K5 serial mode: the streams beside `stream` that batches of PWMs take turns on
  hipEvent_t em_fork = nullptr, em_join[3] = {nullptr, nullptr, nullptr};
  void* d_misc = nullptr;  // small staging buffer
  size_t misc_bytes = 0;
  void* d_bg_partials = nullptr;  // fused K1b: per-block bins
  size_t bg_partials_bytes = 0;
  void* d_iupac_big = nullptr;  // K4 large patterns: bitmap | list | values
  size_t iupac_big_bytes = 0;
  void* d_keys = nullptr;  // partitioned count: bucket regions of 16-bit keys
  size_t keys_bytes = 0;
  void* d_count_aux = nullptr;  // partitioned count: cursors + bucket-major table
  size_t count_aux_bytes = 0;
  uint64_t n_windows_hint = 0;  // total windows of the attached items (0 = unknown: n_items * item_windows)
  uint64_t key_cap_override = 0; // test hook: entries per bucket region (0 = sized from the window count)
  uint64_t iupac_group_bytes = 0; // test hook: scratch budget of one group of large K4 patterns (0 = 1 GiB)
  uint64_t em_table_budget_mb = 0; // K5 serial mode: MiB of weight tables per batch of PWMs (0 = automatic)
  int em_test_skew = 0;         // test hook: every n-th block of the serial EM gets a wrong binade estimate (em.hip, block_binade)
  int em_overlap = 2;           // K5 serial mode, em_serial_scan = 2: streams that batches of PWMs take turns on (1 .. MAX_EM_LANES)
  int em_serial_scan = 2;       // K5 serial mode: cells summed by 3 = the scan of seqsum.h with its blocks evaluated ahead of the
                                // chain, two launches per iteration (W = 10, 12; else as 2), 2 = the same as three launches
                                // (W >= 10; else as 1), 1 = the scan, block after block, 0 = dependent additions
  int em_head_blocks = 1;       // K5 serial mode, W >= 10: the first blocks of every cell folded from zero beside the evaluation of the others
                                // (1 = block 0 only: more measured level at 16 PWMs, slower at 2 and at 1000, profiles/r05_em_kernels.log)
  int sweep_pairs = 1;          // K2+K3, both strands, W >= 12: a pattern and its reverse complement evaluated once (stats_pair_kernel)
  int em_lean_div = 1;          // K5 serial mode: the weights' divisions without range scaling where a PWM's operand ranges allow (em.hip, lean_div)
  int em_fast = 2;              // K5: 2 = the reference's serial float32 sums, bit-exact (default); 1 = one reciprocal per k-mer weight, 0 = the reference's three divisions
  int count_impl = 0;           // 0 auto, 1 direct atomics, 2 partitioned LDS histograms
  int scatter_blocks_per_cu = 0; // tuning hook: workgroups per CU of the partitioned scan (0 = default)
  void* d_sim = nullptr;         // motif similarity grid: PWMs | complements | lengths | sites | scores
  size_t sim_bytes = 0;
  void* comm = nullptr;          // RCCL communicator (comm.hip)
  int comm_transport = 0;        // PENGK_TRANSPORT_*: how the tables of a multi-rank run are exchanged
  int comm_rank = 0, comm_world = 1;
};

enum { PENGK_TRANSPORT_NONE = 0, PENGK_TRANSPORT_RCCL = 1, PENGK_TRANSPORT_TCP = 2 };

namespace pengk {

int fail(int code, const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);
int ensure_scratch(pengk_ctx* ctx, void** slot, size_t* have, size_t need);
int enter(pengk_ctx* ctx);      // hipSetDevice(ctx->device)
int count_init_device();        // per-device kernel attributes of count.hip (current device)
int warm_stats();
int warm_iupac();
int warm_similarity();
int warm_em();  // (one kernel of each translation unit: its code object is loaded)
void comm_release(pengk_ctx* ctx);  // destroys the RCCL communicator, if any

#define PENGK_HIP(call)                                   \
  do {                                                    \
    hipError_t e_ = (call);                               \
    if (e_ != hipSuccess) return ::pengk::hip_fail(e_, #call); \
  } while (0)

inline bool valid_w(int W) { return W >= PENGK_MIN_W && W <= PENGK_MAX_W && (W % 2) == 0; }

// item record fields
constexpr int MAX_EM_LANES = 4;

constexpr uint64_t ITEM_WS_MASK = (1ull << 40) - 1;
constexpr int ITEM_NW_SHIFT = 40;
constexpr uint64_t ITEM_NW_MASK = 0xFFFF;
constexpr int ITEM_CONT_SHIFT = 56;

// Reverse complement of a W-digit little-endian base-4 id by bit arithmetic
// (replaces the half-table lookup of src/base_pattern.cpp:81-97,137-144).
__host__ __device__ inline uint32_t revcomp32(uint32_t id, int W) {
  uint32_t v = ~id;
#if defined(__HIP_DEVICE_COMPILE__)
  v = __brev(v);
#else
  v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
  v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
  v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
  v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
  v = (v >> 16) | (v << 16);
#endif
  v = ((v & 0xAAAAAAAAu) >> 1) | ((v & 0x55555555u) << 1);  // un-reverse the two bits of each digit
  return v >> (32 - 2 * W);
}

// launchers implemented in the .hip files
int launch_count(pengk_ctx* ctx, int both, uint32_t* d_counts, uint64_t* d_ltot, uint64_t* d_bg /* may be NULL */);
int launch_mirror(pengk_ctx* ctx, int W, uint32_t* d_counts);
int launch_bg_count(pengk_ctx* ctx, uint64_t* d_bg);
int launch_bg_model(pengk_ctx* ctx, const uint64_t* d_bg, int K, const float* h_alpha, float* d_V);
int launch_synth(pengk_ctx* ctx, uint64_t seed, uint64_t seq0, uint64_t n_seq, uint32_t L, int W, int item_windows,
                 uint64_t* d_words, uint64_t* d_items);
int launch_stats(pengk_ctx* ctx, int W, int both, int k, int max_k, const float* d_V, const uint64_t* d_ltot,
                 const uint32_t* d_counts, float* d_bgprob, float* d_expected, float* d_logp, float* d_z);
int launch_seed_candidates(pengk_ctx* ctx, int W, const float* d_z, const uint32_t* d_counts, float z_threshold,
                           uint32_t count_threshold, uint32_t cap, uint32_t* d_n, uint32_t* d_ids, float* d_zs);
int launch_sequential_sum(pengk_ctx* ctx, const float* d_terms, uint64_t n_chains, uint64_t chain_len, float* d_out);
int launch_similarity(pengk_ctx* ctx, int n, const float* d_pwm, const float* d_comp, const int32_t* d_len,
                      const uint64_t* d_sites, int both, const float* h_bg, int first_new, float* d_out);
int launch_iupac(pengk_ctx* ctx, int W, int both, const uint64_t* h_ids, int64_t n, const uint32_t* d_counts,
                 const float* d_bgp, const float* d_expected, pengk_iupac_stats* h_out);
int launch_em(pengk_ctx* ctx, int W, int64_t n_pwm, float* d_pwms, float saturation, float threshold, int max_it,
              const uint32_t* d_counts, const float* d_bg, int32_t* d_state, float* d_change);

}  // namespace pengk
