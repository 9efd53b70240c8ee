/* pengk.h -- C ABI of the MI355X (gfx950) hot path of PEnG-motif.
 *
 * The reference (soedinglab/PEnG-motif) has no FFI: its hot path sits behind three C++ classes.
 * This header is the thin extern "C" layer the host-side mirrors of those classes
 * (peng-motif_amd/host/) call; every entry point names the reference function it replaces
 * (file:line relative to the reference tree).  Plain pointers and sizes only; no HIP, torch or
 * C++ types cross the boundary.  All functions return PENGK_OK (0) or a PENGK_ERR_* code and never
 * throw; pengk_last_error() describes the last failure of the calling thread.
 *
 * Pointer naming: `d_` = device memory (hipMalloc'ed by the caller, e.g. through pengk_malloc or a
 * torch tensor's data_ptr()), `h_` = host memory.  Work is enqueued on the context's stream;
 * functions taking only d_ pointers do not synchronise with the host.
 *
 * Encodings (identical to the reference):
 *   pattern id   little-endian base 4, first base = least significant digit (src/base_pattern.h:24-29)
 *   IUPAC id     little-endian base 11, letters A C G T S W R Y M K N = 0..10 (src/iupac_pattern.h:26-29)
 *   BaMM id      big-endian (k+1)-mer id (src/shared/Sequence.cpp:21-33)
 *   V layout     V[0] (4) | V[1] (16) | V[2] (64) floats, 84 in all; bg counts likewise
 *
 * Packed sequence layout in HBM:
 *   words   2 bits per base (A,C,G,T = 0..3), base g in bits [2(g%32), 2(g%32)+2) of 64-bit word g/32;
 *           only the bases of "visited runs" (scan rule of src/base_pattern.cpp:347-381) are stored,
 *           back to back; PENGK_FRONT_PAD_BASES zero bases in front, >= 64 behind.
 *   items   one 64-bit record per scan item: bits 0..39 stream offset of the first window's first
 *           base, bits 40..55 number of windows (1..65535), bit 56 = item continues the previous
 *           item's run (its predecessor windows are stored immediately in front).
 */
#ifndef PENGK_H_
#define PENGK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the library itself is built with -fvisibility=hidden */
#endif

#define PENGK_VERSION 100

#define PENGK_OK 0
#define PENGK_ERR_ARG 1         /* bad argument (NULL, odd/unsupported W, order > 2, ...) */
#define PENGK_ERR_DEVICE 2      /* no usable gfx950 device / HIP runtime error */
#define PENGK_ERR_RANGE 3       /* a 32-bit count bin could overflow on this shard */
#define PENGK_ERR_UNSUPPORTED 4 /* operation not defined for this input (see function) */
#define PENGK_ERR_NOMEM 5

#define PENGK_MIN_W 2 /* the reference accepts every even pattern length (src/Global.cpp:103-106) */
#define PENGK_MAX_W 14
#define PENGK_FRONT_PAD_BASES 64
#define PENGK_DEFAULT_ITEM_WINDOWS 256
#define PENGK_MIN_ITEM_WINDOWS 64

typedef struct pengk_ctx pengk_ctx;

/* ---- lifecycle, errors, device memory ------------------------------------------------------- */
int pengk_version(void);
const char* pengk_last_error(void);
const char* pengk_error_name(int code);

/* Binds a context to HIP device `device` and creates its stream.  Fails with PENGK_ERR_DEVICE when
 * no GPU is present -- there is no CPU fallback in this library. */
int pengk_create(int device, pengk_ctx** out);
int pengk_destroy(pengk_ctx* ctx);
int pengk_synchronize(pengk_ctx* ctx);
/* hipStream_t of the context (for callers that record their own events). */
void* pengk_stream(pengk_ctx* ctx);
/* Re-target the context to an externally owned hipStream_t (e.g. torch's current stream). */
int pengk_set_stream(pengk_ctx* ctx, void* hip_stream);

/* Tunables / introspection.  Options: "count_impl" 0 = auto, 1 = direct global atomics, 2 = partitioned LDS
 * histograms (W = 8 .. 14); "n_windows_hint" = total windows of the attached items (sizes the key buffer
 * tightly; set it after pengk_set_sequences); "key_cap_override" (test hook) entries per bucket region of the
 * partitioned count, 0 = automatic; "sweep_pairs" 1 (default) / 0: both strands from W = 12 on, a pattern and its reverse complement evaluated once (0: one thread per pattern; same bits); "iupac_group_bytes" (test hook) scratch budget for one group of large
 * patterns in pengk_iupac_aggregate, 0 = 1 GiB; "em_fast" 2 (default) / 1 / 0, see pengk_em.  Info: "deferred_items" (of the last pengk_count;
 * synchronises), "num_cu"; of the last pengk_em / pengk_em_device call in the serial mode with its blocks evaluated ahead
 * (synchronise): "em_fetched_blocks" (blocks a chain added term by term), "em_mispredicted_blocks" (of those: blocks
 * whose estimated binade did not hold), "em_restaged_blocks" / "em_restaged_waits" (csrc/seqsum.h, WalkCounts). */
int pengk_set_option(pengk_ctx* ctx, const char* name, int64_t value);
int pengk_get_info(pengk_ctx* ctx, const char* name, int64_t* value_out);

int pengk_malloc(pengk_ctx* ctx, size_t bytes, void** d_out);
int pengk_free(pengk_ctx* ctx, void* d_ptr);
int pengk_memcpy_h2d(pengk_ctx* ctx, void* d_dst, const void* h_src, size_t bytes); /* synchronous */
int pengk_memcpy_d2h(pengk_ctx* ctx, void* h_dst, const void* d_src, size_t bytes); /* synchronous */
/* (No counterpart in the reference, whose process has no device runtime to start.)
 * Pays the process's first-use costs ahead of time (first device-to-host copy: 10-17 ms whatever its size; the code
 * objects of the sweep / IUPAC / EM / similarity kernels).  Optional; call once after pengk_create, from any thread,
 * beside other work. */
int pengk_warmup(pengk_ctx* ctx);
int pengk_memset(pengk_ctx* ctx, void* d_dst, int byte, size_t bytes);              /* async */
/* Page-locked host memory for result tables: the reference's BasePattern hands out raw host arrays
 * (src/base_pattern.h:129-140: size_t counts, float probabilities / expected / z / log-p, 4^W each); mirrors of
 * the device tables land in them at link speed when they are allocated here.  Ordinary host pointers otherwise. */
int pengk_host_alloc(pengk_ctx* ctx, size_t bytes, void** h_out);
int pengk_host_free(pengk_ctx* ctx, void* h_ptr);

/* Stream-ordered event timing without exposing HIP types: records an event on the context's stream;
 * pengk_timer_elapsed_ms synchronises on `stop`. */
int pengk_timer_create(pengk_ctx* ctx, void** timer_out);
int pengk_timer_record(pengk_ctx* ctx, void* timer);
int pengk_timer_elapsed_ms(pengk_ctx* ctx, void* start, void* stop, float* ms_out);
int pengk_timer_destroy(pengk_ctx* ctx, void* timer);

/* ---- host packer (pure CPU; replaces the per-sequence byte codes + revcomp copies that
 *      count_patterns walks, src/base_pattern.cpp:339-342, src/shared/Sequence.cpp:4-35) -------- */
typedef struct pengk_packed {
  uint64_t* words;       /* 2-bit stream incl. padding */
  uint64_t n_words;
  uint64_t* items;       /* scan items */
  uint64_t n_items;
  uint64_t n_bases;      /* stored bases (without padding) */
  uint64_t n_windows;    /* = ltot: number of visited windows (src/base_pattern.cpp:367) */
  uint64_t max_bin_bound;/* upper bound on any single count bin: sum over runs of ceil(windows/W) */
  int64_t bg_counts[84]; /* (k+1)-mer counts, k = 0..2, exactly as BackgroundModel counts them incl. the
                            invalid-base rule (src/shared/BackgroundModel.cpp:60-84) */
  uint64_t n_sequences;
  uint64_t max_len;      /* longest input sequence */
  int W;
  int item_windows;
  int all_whole;         /* 1 iff every input sequence is exactly one stored run (no invalid base, L >= W):
                            then pengk_bg_count can recount bg_counts on the device */
} pengk_packed;

/* codes: the reference's byte codes (0 = other, A,C,G,T = 1..4) of all sequences back to back;
 * offs: n_seq+1 offsets into codes.  item_windows: maximum windows per scan item
 * (>= PENGK_MIN_ITEM_WINDOWS, <= 65535; 0 = PENGK_DEFAULT_ITEM_WINDOWS).
 * out->words / out->items are owned by the library (zero-filled blocks on transparent huge pages where the kernel
 * grants them): release them with pengk_packed_free, never with free(). */
int pengk_pack(const uint8_t* h_codes, const int64_t* h_offs, int64_t n_seq, int W, int item_windows,
               pengk_packed* out);
/* The same with a given number of host threads (0 = automatic, as pengk_pack): a caller that packs many chunks of an
 * input from its own worker threads -- the CLI does, while the FASTA file is still being read -- asks for 1. */
int pengk_pack_threads(const uint8_t* h_codes, const int64_t* h_offs, int64_t n_seq, int W, int item_windows, int threads,
                       pengk_packed* out);
void pengk_packed_free(pengk_packed* p);

/* Packing an input chunk by chunk into ONE pair of caller-owned host buffers (zero-filled; e.g. fresh anonymous
 * memory), from any number of threads at once: every call reserves its place with the two cursors (atomically), writes
 * its words -- own zero padding in front and behind, like pengk_pack -- and its items with ABSOLUTE stream offsets, so
 * that words[0 .. word_cursor) / items[0 .. item_cursor) are attached to the device as they stand, with ONE
 * pengk_set_sequences: max_bin_bound = the sum over the chunks, all_whole = the AND.  Counts are additive over
 * sequences, so the order in which chunks land does not matter.  `out` receives the chunk's figures (windows, bounds,
 * background counters ...); out->words / out->items point INTO the buffers and are not to be released.
 * The CLI packs every chunk of the FASTA file this way while the rest is still being read (host/device.cpp).
 * Capacity: words >= sum over chunks of (64 + bases + 31) / 32 + 4, items >= their number; PENGK_ERR_RANGE otherwise. */
typedef struct pengk_pack_target {
  uint64_t* words;
  uint64_t words_cap;
  uint64_t* items;
  uint64_t items_cap;
  uint64_t word_cursor; /* next free word / item: start at 0; advanced atomically by the calls */
  uint64_t item_cursor;
} pengk_pack_target;
int pengk_pack_append(const uint8_t* h_codes, const int64_t* h_offs, int64_t n_seq, int W, int item_windows,
                      pengk_pack_target* target, pengk_packed* out);

/* ---- device-resident sequences ------------------------------------------------------------- */
/* Attach caller-owned device buffers holding a packed stream and its items (layout above).
 * `all_whole` as in pengk_packed.  The buffers must outlive every later call that scans them. */
int pengk_set_sequences(pengk_ctx* ctx, const uint64_t* d_words, uint64_t n_words, const uint64_t* d_items,
                        uint64_t n_items, int W, int item_windows, uint64_t max_bin_bound, int all_whole);

/* Size query + on-device generation of the synthetic input of SURVEY.md 8d (sequences
 * [seq0, seq0+n_seq) of length L, splitmix64 counter-based, motif GCTGAGTCAT planted in 10 %),
 * written straight into caller-owned device buffers in the packed layout, then attached as by
 * pengk_set_sequences.  Bit-identical to the CPU generator the tests use. */
int pengk_synth_sizes(uint64_t n_seq, uint32_t L, int W, int item_windows, uint64_t* n_words, uint64_t* n_items);
int pengk_synth_sequences(pengk_ctx* ctx, uint64_t seed, uint64_t seq0, uint64_t n_seq, uint32_t L, int W,
                          int item_windows, uint64_t* d_words, uint64_t* d_items);

/* ---- K1: 4^W k-mer count (BasePattern::count_patterns / count_patterns_single_strand,
 *      src/base_pattern.cpp:331-441) ---------------------------------------------------------------
 * d_counts: uint32[4^W], overwritten.  both_strands: counts land on the canonical id min(id, rc)
 * only (run pengk_mirror_counts for the reference's twin copy, :387-392).  d_ltot: uint64 scalar,
 * overwritten with the number of visited windows.  The non-overlap rule (:361-366) is applied
 * exactly, with 64-bit positions.  Counts of different shards add (the rule is per sequence).
 * Emitters: W = 8 .. 14 use the partitioned LDS-histogram count (one partition level at W = 8, 10; two at W = 12; three
 * at W = 14: 2^28 bins = 2^13 LDS histograms -- the reference itself advises W <= 12, README.md:119); W = 4, 6 (tables
 * that are contended in any form) count with one device-scope atomic per window. */
int pengk_count(pengk_ctx* ctx, int both_strands, uint32_t* d_counts, uint64_t* d_ltot);
int pengk_mirror_counts(pengk_ctx* ctx, int W, uint32_t* d_counts);
/* pengk_count with K1b fused into the same scan (the rolling id's top three digits are the 3-mer ending at
 * the current base): additionally writes uint64[84] background counts as pengk_bg_count would.  Same
 * restriction as pengk_bg_count (whole-sequence runs), else PENGK_ERR_UNSUPPORTED. */
int pengk_count_bg(pengk_ctx* ctx, int both_strands, uint32_t* d_counts, uint64_t* d_ltot, uint64_t* d_bg_counts);

/* K1b: background (k+1)-mer counts k = 0..2 (BackgroundModel ctor loop,
 * src/shared/BackgroundModel.cpp:60-84) -> uint64[84].  Only for inputs whose sequences are all
 * whole runs (pengk_packed.all_whole); otherwise PENGK_ERR_UNSUPPORTED and the packer's
 * bg_counts are the source. */
int pengk_bg_count(pengk_ctx* ctx, uint64_t* d_bg_counts);

/* BackgroundModel::calculateV (src/shared/BackgroundModel.cpp:490-530), interpolated, on device:
 * uint64[84] counts -> float[84] conditional probabilities.  alpha: 3 host floats. */
int pengk_bg_model(pengk_ctx* ctx, const uint64_t* d_bg_counts, int K, const float* h_alpha, float* d_V);

/* ---- K2+K3: pattern-space sweep (calculate_bg_probabilities + aggregate_double_strand_background +
 *      calculate_expected_counts + calculate_log_pvalues + calculate_zscores,
 *      src/base_pattern.cpp:231-325) ---------------------------------------------------------------
 * d_bgprob: float[(max_k+1)][4^W] (order-major), strand-aggregated when both_strands.
 * expected/z/logp use order k <= max_k <= 2.  d_counts must already be mirrored for both strands.
 * d_ltot: the uint64 scalar pengk_count wrote (after any cross-GPU reduction). */
int pengk_pattern_stats(pengk_ctx* ctx, int W, int both_strands, int k, int max_k, const float* d_V,
                        const uint64_t* d_ltot, const uint32_t* d_counts, float* d_bgprob, float* d_expected,
                        float* d_logp, float* d_z);

/* ---- seed candidates (first half of BasePattern::select_base_patterns, src/base_pattern.cpp:443-515) ----------------
 * The reference std::sorts all 4^W ids by z-score and walks the ranking down to the threshold (:458-466); only ids with
 * z >= z_threshold and count >= count_threshold can become seeds.  This call compacts exactly those on the device
 * (unordered) into h_ids / h_z (capacity entries each); *n_out = their number -- when it exceeds capacity only the
 * first `capacity` found were copied and the caller retries with room for n_out.  The caller ranks the survivors.
 * NOTE the ranking of EXACT z ties (every reverse-complement pair under both strands) is std::sort's in the reference,
 * i.e. unspecified: a caller that ranks by (z descending, id ascending) reports the same seed set up to the strand a
 * pair is named on.  The host mirror keeps the reference's ranking by default (host/ranked_prefix.h) and uses this
 * call only on request (PENGK_SEED_SELECT=device). */
int pengk_seed_candidates(pengk_ctx* ctx, int W, const float* d_z, const uint32_t* d_counts, float z_threshold,
                          uint64_t count_threshold, uint32_t* h_ids, float* h_z, int64_t capacity, int64_t* n_out);

/* ---- K4: IUPAC degenerate-pattern aggregation (IUPACPattern::aggregate_attributes_from_basepatterns
 *      and count_combined_occurences, src/iupac_pattern.cpp:331-473,806-833) -------------------- */
typedef struct pengk_iupac_stats {
  uint64_t sites;   /* sum of counts over the distinct underlying k-mers */
  float bg_p;       /* float32 sum of background probabilities, in the reference's summation order */
  float expected;   /* float32 sum of expected counts, same order */
  float zscore;     /* :446 */
  float log_pvalue; /* :453-470, Bonferroni term included */
} pengk_iupac_stats;

/* n patterns (host ids) -> n results (host).  d_bgp: the order-k table used for z-scores. */
int pengk_iupac_aggregate(pengk_ctx* ctx, int W, int both_strands, const uint64_t* h_iupac_ids, int64_t n,
                          const uint32_t* d_counts, const float* d_bgp, const float* d_expected,
                          pengk_iupac_stats* h_out);

/* ---- K5: EM over the whole 4^W table (Peng::em_optimize_pwms + calculate_prob_odds,
 *      src/peng.cpp:48-197; row normalisation src/iupac_pattern.cpp:291-303) ----------------------
 * h_pwms: n_pwm x W x 4 floats, updated in place with the PWM the reference's loop ends on (before the
 * extra normalisation of the IUPACPattern(ori, pwm) constructor).  Three modes (option "em_fast"):
 *   2  (library default) serial: the reference's float32 arithmetic including the ORDER in which it adds the 4^W weights of a PWM
 *      cell (src/peng.cpp:121-127) -- PWMs, iteration counts and `change` are the reference's bit for bit.  From
 *      W = 8 on a cell's chain of roundings is evaluated as a scan (csrc/seqsum.h; a PWM with a negative or non-finite
 *      weight is summed by a plain loop).  What a caller needs when discrete decisions follow (motif merging compares
 *      similarity scores that are exactly tied in real arithmetic for reverse-complement twins); the CLI's default.
 *      Options of this mode (pengk_set_option):
 *      How the sum is carried out: W >= 10 -- the scan with its blocks of 4096 terms evaluated AHEAD of the chain: all
 *      blocks of all cells at once, each under the binade a prefix of plain block sums predicts for it, then one addition
 *      per block along the chain (three launches per iteration; the previous iteration's finalize step sits at the head of
 *      the first); W = 8 -- the scan, block after block; W <= 6 -- one dependent addition after the other (csrc/em_legacy.hip).
 *        "em_serial_scan"      2 (default) as above; 3 (W = 10, 12) the first two launches of an iteration as ONE kernel that
 *                              keeps a span's weights in LDS (half the table traffic at W = 12; measured no faster).
 *        "em_overlap"          W >= 10: streams the batches of PWMs take turns on, 1..4 (default 2; the
 *                              context's own stream waits for the others before the call returns or copies).
 *        "em_head_blocks"      W >= 10, 1..64 (default 1): the first blocks of every cell -- where the sum doubles from
 *                              block to block -- are folded from zero beside the evaluation of the others, the chain
 *                              starts behind them.  Results do not depend on it (and more than block 0 measured no faster).
 *        "em_test_skew"        (test hook) n > 0: about every n-th block gets a WRONG binade estimate -- results must not
 *                              change, only the time (the estimate never carries exactness).  0 = off.
 *        "em_test_lookback"    (test hook, "em_serial_scan" = 3) n > 0: every n-th workgroup acts as if the look-back for
 *                              its estimates had timed out (its blocks are then folded by their chains).  0 = off.
 *        "em_table_budget_mb"  MiB of weight tables in flight (4^W floats per PWM; twice that where the W = 8 scan keeps a
 *                              second copy in position 0's order); the PWMs of a call go
 *                              through in batches of that size.  0 (default) = automatic: 192 MiB for W <= 10 (what a
 *                              batch writes is still in the 256 MiB Infinity Cache when it is read), above that a
 *                              quarter of the free memory, at most 24 GiB.
 *   0  the reference's float32 terms (three divisions per k-mer weight), summed in fp64 through a fixed tree.
 *   1  (opt-in) the weight c*s / (1 + s/(prod/bg)) evaluated as c*s*prod / (prod + s*bg) with one
 *      reciprocal (~1 ulp per term), fp64 tree sums: the throughput mode, 2.3e12 PWM-k-mer evaluations/s.
 * Modes 0 and 1 agree with the reference within BASELINE.json's 1e-5 relative (the reference's own serial float32
 * sums are off by up to 2.6e-4 relative from the exact ones); the float32 product over the PWM columns is built in
 * the reference's order in all modes.
 * max_iterations <= 0: no iteration (the reference's loop condition, src/peng.cpp:104), the PWMs come back unchanged.
 * h_iters / h_change (optional): iterations run and last `change` per PWM. */
int pengk_em(pengk_ctx* ctx, int W, int64_t n_pwm, float* h_pwms, float saturation, float threshold,
             int max_iterations, const uint32_t* d_counts, const float* d_bg, int* h_iters, float* h_change);

/* Test hook: the serial mode (em_fast = 2) by an EARLIER GENERATION of this library, for cross-checks against the
 * current one (csrc/em_legacy.hip): 0 = one dependent addition after the other, 1 = the scan of csrc/seqsum.h block after
 * block (needs W >= 8; below: as 0); 2 / 3 = back to the library's scheme ("em_serial_scan").  Every generation returns
 * the reference's sums bit for bit; only the time differs. */
int pengk_test_em_generation(pengk_ctx* ctx, int generation);

/* Device-resident variant for benchmarking / pipelines: d_pwms n_pwm x W x 4 floats in HBM, updated in
 * place; d_state: n_pwm x 2 int32 scratch {iterations, active}; no host synchronisation. */
int pengk_em_device(pengk_ctx* ctx, int W, int64_t n_pwm, float* d_pwms, float saturation, float threshold,
                    int max_iterations, const uint32_t* d_counts, const float* d_bg, int32_t* d_state,
                    float* d_change);

/* ---- sequential float32 sums (the summation order of src/peng.cpp:121-127 on plain arrays) ---------------------
 * d_out[i] = ((0 + t[0]) + t[1]) + ... over the chain_len floats at d_terms + i * chain_len, every addition rounded
 * to float32 like a left-to-right CPU loop: bit-exact, including denormals and overflow to +inf.  Chains of
 * non-negative finite terms are evaluated by the scan of csrc/seqsum.h (one wave per chain, 4096 terms per step);
 * a chain with a negative, infinite or NaN term is summed by a plain loop on the device.  Device pointers. */
int pengk_sequential_sum_f32(pengk_ctx* ctx, const float* d_terms, uint64_t n_chains, uint64_t chain_len, float* d_out);

/* ---- motif similarity grid (Peng::merge_iupac_patterns' inner loops, src/peng.cpp:251-272, over
 *      IUPACPattern::calculate_S, src/iupac_pattern.cpp:568-615) -------------------------------------------------
 * n motifs: h_pwm / h_comp = n x PENGK_MAX_MOTIF_LEN x 4 floats (PWM and its reverse-complement PWM, rows beyond
 * h_len[i] unused), h_sites[i] = the motif's site count (decides which of a pair is complemented, :592-597), h_bg = the
 * 4 background letter frequencies.  For every pair (i, j), i < j, j >= first_new -- ordered j = first_new .. n-1, then
 * i = 0 .. j-1 -- h_out receives max over strands (both_strands) and shifts with >= 6 overlapping columns of the score
 * s (calculate_s, :551-566), -inf when no shift qualifies.  first_new = 0: the whole triangle; first_new = n - 1: the
 * new motif against all others after a merge.
 * The values are an fp64 evaluation rounded to float; the reference rounds every partial sum to float, so they agree
 * to ~1e-4, NOT bit for bit.  Use: find the pairs that can be the maximum, evaluate those with the reference's
 * arithmetic (the host mirror does exactly that with a margin of 2e-3). */
#define PENGK_MAX_MOTIF_LEN 64
int pengk_motif_similarity(pengk_ctx* ctx, int n, const float* h_pwm, const float* h_comp, const int32_t* h_len,
                           const uint64_t* h_sites, int both_strands, const float* h_bg, int first_new, float* h_out);

/* Self-test of the division sequence the serial EM's weights kernel uses where a PWM's operand ranges allow (the IEEE
 * division's instructions without its range scaling: csrc/em.hip, lean_div; src/peng.cpp:124-125, 186 are the three
 * divisions of a weight).  4096 x 256 threads draw pairs_per_thread random operand pairs each, keep those inside the
 * guard's domain and compare with the compiler's division bit for bit.  h_out[0] = pairs compared, [1] = pairs that
 * differ (must be 0), [2] = of the compared: pairs with a numerator of zero. */
int pengk_selftest_division(pengk_ctx* ctx, uint64_t seed, uint32_t pairs_per_thread, uint64_t* h_out /* [3] */);

/* ---- C1: the one exchange step of a multi-GPU run (no counterpart in the reference, which is a single
 *      process).  One process per GPU; sequences shard by whole records; every rank counts its shard, then the
 *      count table, ltot and the 84 background counters are summed over the ranks -- exact, because the non-overlap
 *      rule never crosses a sequence boundary (src/base_pattern.cpp:382) -- by ONE grouped RCCL all-reduce over xGMI on
 *      the context's stream.  Afterwards every rank holds the global tables: the sweeps are replicated, the EM splits
 *      the PWM list, pengk_allgather returns the pieces.  librccl is opened on first use.
 *
 *      Rendezvous: either the caller distributes rank 0's id itself (pengk_comm_unique_id + pengk_comm_init, e.g. over
 *      an existing launcher's store), or pengk_comm_init_env reads RANK / WORLD_SIZE / MASTER_ADDR from the launcher
 *      environment (torchrun, mpirun wrappers), opens the process's host channel (below) and hands the id out over it.
 *      With WORLD_SIZE unset or 1 every call below is a no-op that succeeds.
 *
 *      Host channel: what the HOST side of a sharded run has to agree on before any table exists -- every rank reads
 *      only its byte range of the FASTA file (replaces the whole-file pass of src/shared/SequenceSet.cpp:285-447 per
 *      rank), so the number of records, the base counts, the 84 background counters
 *      (src/shared/BackgroundModel.cpp:60-84 -- additive in the counters, not in V) and the reader's warnings are
 *      combined over a star of TCP connections to rank 0 on PENGK_COMM_PORT (default MASTER_PORT + 17), bound to
 *      MASTER_ADDR.  Every socket operation has a deadline of PENGK_COMM_TIMEOUT seconds (default 120) and fails with
 *      PENGK_ERR_DEVICE when a rank is missing; peers are admitted only with their rank and a token derived from the
 *      launcher environment (plus PENGK_COMM_TOKEN, if set).  A rank that fails on its own must exit non-zero: its
 *      peers then fail at their next host-channel call, or are torn down by the launcher.
 *
 *      PENGK_COMM_TRANSPORT=tcp makes pengk_comm_init_env skip RCCL and exchange the tables through host memory over
 *      the host channel: a rehearsal transport for boxes with fewer GPUs than ranks (RCCL cannot put two ranks on one
 *      GPU), never selected automatically. */
int pengk_comm_host_init_env(void);                 /* idempotent; collective over the ranks of the job */
int pengk_comm_host_info(int* rank_out, int* world_out);
int pengk_comm_host_allgather(const void* h_send, void* h_recv, size_t bytes_per_rank);
int pengk_comm_host_allreduce_u64(uint64_t* h_buf, size_t n); /* in-place sum */
int pengk_comm_host_shutdown(void);

#define PENGK_COMM_ID_BYTES 128
int pengk_comm_unique_id(void* id_out /* PENGK_COMM_ID_BYTES */);
int pengk_comm_init(pengk_ctx* ctx, const void* id, int rank, int world);
int pengk_comm_init_env(pengk_ctx* ctx);
int pengk_comm_info(pengk_ctx* ctx, int* rank_out, int* world_out);
/* 1 after a pengk_comm_init / pengk_comm_init_env whose deadline passed: a helper thread of this process is then still inside
 * ncclCommInitRank (which has no deadline of its own, and no communicator yet that ncclCommAbort could be given).  Such a
 * process must leave with _exit(): exit handlers and static destructors of HIP / RCCL under that live thread can crash or
 * hang.  (peng_motif does; csrc/comm.hip.) */
int pengk_comm_init_abandoned(void);
/* ncclGetVersion's code of the librccl this process bound (0 before the first communicator call); libraries that report
 * an API older than NCCL 2.0 are refused when they are loaded. */
int pengk_comm_rccl_version(void);
int pengk_comm_destroy(pengk_ctx* ctx);
/* In-place sum over the ranks of d_counts uint32[4^W] (the global bin bound must stay below 2^32: the caller checks
 * the sum of its shards' pengk_packed.max_bin_bound), d_ltot uint64[1] and, if not NULL, d_bg uint64[84]. */
int pengk_allreduce_tables(pengk_ctx* ctx, int W, uint32_t* d_counts, uint64_t* d_ltot, uint64_t* d_bg_counts);
/* Sums the attached shards' bin bounds (pengk_set_sequences' max_bin_bound) over the ranks and fails with
 * PENGK_ERR_RANGE when a 32-bit bin could overflow globally.  Collective; synchronises with the host. */
int pengk_comm_check_bin_bound(pengk_ctx* ctx);
/* d_recv[r * bytes_per_rank ...) = rank r's d_send[0 .. bytes_per_rank), for every r (ncclAllGather). */
int pengk_allgather(pengk_ctx* ctx, const void* d_send, void* d_recv, size_t bytes_per_rank);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* PENGK_H_ */
