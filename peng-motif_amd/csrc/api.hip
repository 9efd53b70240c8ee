// api.hip -- context, device memory, timers and argument checking of the pengk C ABI.
// The kernels live in count.hip / stats.hip / iupac.hip / em.hip; the host packer in pack.cpp.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <time.h>

#include <new>

#include "pengk_internal.h"

namespace pengk {

static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

int hip_fail(hipError_t e, const char* what) {
  return fail(PENGK_ERR_DEVICE, "%s: %s", what, hipGetErrorString(e));
}

int ensure_scratch(pengk_ctx* ctx, void** slot, size_t* have, size_t need) {
  if (*have >= need) return PENGK_OK;
  if (*slot) PENGK_HIP(hipFree(*slot));
  *slot = nullptr;
  *have = 0;
  PENGK_HIP(hipMalloc(slot, need));
  *have = need;
  return PENGK_OK;
}

// Every entry point that launches, allocates or copies runs on the context's device, whatever device the calling
// thread had current (the CLI creates the context on a helper thread and uses it from the main thread).
int enter(pengk_ctx* ctx) {
  PENGK_HIP(hipSetDevice(ctx->device));
  return PENGK_OK;
}

static int warm_code_objects() {
  int rc = warm_stats();
  if (!rc) rc = warm_iupac();
  if (!rc) rc = warm_em();
  if (!rc) rc = warm_similarity();
  return rc;
}

}  // namespace pengk

using namespace pengk;

#define PENGK_ENTER(ctx)               \
  do {                                 \
    int rc_enter_ = ::pengk::enter(ctx); \
    if (rc_enter_) return rc_enter_;   \
  } while (0)

extern "C" {

int pengk_version(void) { return PENGK_VERSION; }
const char* pengk_last_error(void) { return g_err; }

const char* pengk_error_name(int code) {
  switch (code) {
    case PENGK_OK: return "PENGK_OK";
    case PENGK_ERR_ARG: return "PENGK_ERR_ARG";
    case PENGK_ERR_DEVICE: return "PENGK_ERR_DEVICE";
    case PENGK_ERR_RANGE: return "PENGK_ERR_RANGE";
    case PENGK_ERR_UNSUPPORTED: return "PENGK_ERR_UNSUPPORTED";
    case PENGK_ERR_NOMEM: return "PENGK_ERR_NOMEM";
    default: return "PENGK_ERR_?";
  }
}

int pengk_create(int device, pengk_ctx** out) {
  if (!out) return fail(PENGK_ERR_ARG, "pengk_create: out is NULL");
  *out = nullptr;
  // PENGK_TIMING_CREATE=1: where the start-up of the runtime goes (stderr)
  const bool timing = getenv("PENGK_TIMING_CREATE") != nullptr;
  timespec t_prev;
  clock_gettime(CLOCK_MONOTONIC, &t_prev);
  auto lap = [&](const char* what) {
    if (!timing) return;
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    fprintf(stderr, "[pengk_create] %s: %.1f ms\n", what, (t.tv_sec - t_prev.tv_sec) * 1e3 + (t.tv_nsec - t_prev.tv_nsec) * 1e-6);
    t_prev = t;
  };
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  lap("hipGetDeviceCount (runtime start)");
  if (e != hipSuccess || n <= 0)
    return fail(PENGK_ERR_DEVICE, "pengk_create: no HIP device (%s); this library has no CPU fallback",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (device < 0 || device >= n) return fail(PENGK_ERR_ARG, "pengk_create: device %d out of range [0,%d)", device, n);
  PENGK_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  PENGK_HIP(hipGetDeviceProperties(&prop, device));
  lap("hipSetDevice + properties");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(PENGK_ERR_DEVICE, "pengk_create: device %d is %s; this build targets gfx950 only", device, prop.gcnArchName);
  pengk_ctx* c = new (std::nothrow) pengk_ctx();
  if (!c) return fail(PENGK_ERR_NOMEM, "pengk_create: out of host memory");
  c->device = device;
  c->num_cu = prop.multiProcessorCount;
  e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete c;
    return hip_fail(e, "hipStreamCreate");
  }
  c->own_stream = true;
  lap("stream");
  int rc = count_init_device();  // per-device kernel attributes (128 KiB dynamic LDS of pass B)
  lap("kernel attributes (code object load)");
  if (rc) {
    (void)hipStreamDestroy(c->stream);
    delete c;
    return rc;
  }
  *out = c;
  return PENGK_OK;
}

int pengk_destroy(pengk_ctx* ctx) {
  if (!ctx) return PENGK_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  comm_release(ctx);
  if (ctx->d_defer) (void)hipFree(ctx->d_defer);
  if (ctx->d_em_partials) (void)hipFree(ctx->d_em_partials);
  if (ctx->d_em_tables) (void)hipFree(ctx->d_em_tables);
  if (ctx->d_em_blocks) (void)hipFree(ctx->d_em_blocks);
  if (ctx->d_em_counters) (void)hipFree(ctx->d_em_counters);
  if (ctx->d_em_look) (void)hipFree(ctx->d_em_look);
  if (ctx->d_pair_mids) (void)hipFree(ctx->d_pair_mids);
  for (int l = 0; l < 3; ++l) {
    if (ctx->em_streams[l]) (void)hipStreamDestroy(ctx->em_streams[l]);
    if (ctx->em_join[l]) (void)hipEventDestroy(ctx->em_join[l]);
  }
  if (ctx->em_fork) (void)hipEventDestroy(ctx->em_fork);
  if (ctx->d_misc) (void)hipFree(ctx->d_misc);
  if (ctx->d_keys) (void)hipFree(ctx->d_keys);
  if (ctx->d_iupac_big) (void)hipFree(ctx->d_iupac_big);
  if (ctx->d_bg_partials) (void)hipFree(ctx->d_bg_partials);
  if (ctx->d_count_aux) (void)hipFree(ctx->d_count_aux);
  if (ctx->d_sim) (void)hipFree(ctx->d_sim);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return PENGK_OK;
}

int pengk_synchronize(pengk_ctx* ctx) {
  if (!ctx) return fail(PENGK_ERR_ARG, "ctx is NULL");
  PENGK_ENTER(ctx);
  PENGK_HIP(hipStreamSynchronize(ctx->stream));
  return PENGK_OK;
}

int pengk_set_option(pengk_ctx* ctx, const char* name, int64_t value) {
  if (!ctx || !name) return fail(PENGK_ERR_ARG, "pengk_set_option: NULL argument");
  if (strcmp(name, "count_impl") == 0) {
    if (value < 0 || value > 2) return fail(PENGK_ERR_ARG, "count_impl must be 0 (auto), 1 (direct) or 2 (partition)");
    ctx->count_impl = (int)value;
    return PENGK_OK;
  }
  if (strcmp(name, "key_cap_override") == 0) {
    ctx->key_cap_override = (uint64_t)value;
    return PENGK_OK;
  }
  if (strcmp(name, "em_fast") == 0) {
    if (value < 0 || value > 2) return fail(PENGK_ERR_ARG, "em_fast must be 0 (reference terms), 1 (fast) or 2 (serial, bit-exact)");
    ctx->em_fast = (int)value;
    return PENGK_OK;
  }
  if (strcmp(name, "em_serial_scan") == 0) {
    if (value < 2 || value > 3)
      return fail(PENGK_ERR_ARG, "em_serial_scan must be 2 (blocks evaluated ahead of their chain, three launches per iteration) or 3 (two: weights and block evaluation as one kernel); earlier generations: pengk_test_em_generation");
    ctx->em_serial_scan = (int)value;
    return PENGK_OK;
  }
  if (strcmp(name, "em_test_lookback") == 0) {
    if (value < 0 || value > 1000000) return fail(PENGK_ERR_ARG, "em_test_lookback must be 0 (off) or the n of 'every n-th workgroup'");
    ctx->em_test_lookback = (int)value;
    return PENGK_OK;
  }
  if (strcmp(name, "em_test_skew") == 0) {
    if (value < 0 || value > 1000000) return fail(PENGK_ERR_ARG, "em_test_skew must be 0 (off) or the n of 'every n-th block'");
    ctx->em_test_skew = (int)value;
    return PENGK_OK;
  }
  if (strcmp(name, "em_head_blocks") == 0) {
    if (value < 1 || value > 64) return PENGK_ERR_ARG;
    ctx->em_head_blocks = (int)value;
    return PENGK_OK;
  }
  if (strcmp(name, "sweep_pairs") == 0) {
    ctx->sweep_pairs = value != 0;
    return PENGK_OK;
  }
  if (strcmp(name, "em_lean_div") == 0) {
    ctx->em_lean_div = value != 0;
    return PENGK_OK;
  }
  if (strcmp(name, "em_overlap") == 0) {
    if (value < 1 || value > MAX_EM_LANES) return fail(PENGK_ERR_ARG, "em_overlap must be 1 .. %d streams", MAX_EM_LANES);
    ctx->em_overlap = (int)value;
    return PENGK_OK;
  }
  if (strcmp(name, "em_table_budget_mb") == 0) {
    if (value < 0) return fail(PENGK_ERR_ARG, "em_table_budget_mb must be >= 0 (0 = automatic)");
    ctx->em_table_budget_mb = (uint64_t)value;
    return PENGK_OK;
  }
  if (strcmp(name, "iupac_group_bytes") == 0) {
    ctx->iupac_group_bytes = (uint64_t)value;
    return PENGK_OK;
  }
  if (strcmp(name, "scatter_blocks_per_cu") == 0) {
    if (value < 0 || value > 8) return fail(PENGK_ERR_ARG, "scatter_blocks_per_cu must be 0 (default) .. 8");
    ctx->scatter_blocks_per_cu = (int)value;
    return PENGK_OK;
  }
  if (strcmp(name, "n_windows_hint") == 0) {
    ctx->n_windows_hint = (uint64_t)value;
    return PENGK_OK;
  }
  return fail(PENGK_ERR_ARG, "unknown option '%s'", name);
}

int pengk_get_info(pengk_ctx* ctx, const char* name, int64_t* value) {
  if (!ctx || !name || !value) return fail(PENGK_ERR_ARG, "pengk_get_info: NULL argument");
  PENGK_ENTER(ctx);
  if (strcmp(name, "deferred_items") == 0) {  // scan items the last pengk_count handed to the exact fallback
    uint32_t n = 0;
    if (ctx->d_defer) {
      PENGK_HIP(hipMemcpyAsync(&n, ctx->d_defer, sizeof n, hipMemcpyDeviceToHost, ctx->stream));
      PENGK_HIP(hipStreamSynchronize(ctx->stream));
    }
    *value = n;
    return PENGK_OK;
  }
  if (strcmp(name, "num_cu") == 0) {
    *value = ctx->num_cu;
    return PENGK_OK;
  }
  // K5 serial mode, blocks ahead of their chain: what the chains of the LAST pengk_em / pengk_em_device call met, summed
  // over all cells, PWMs and iterations (seqsum::WalkCounts; 0 for every other EM mode).  Synchronises with the stream.
  static const char* const em_names[4] = {"em_fetched_blocks", "em_mispredicted_blocks", "em_restaged_blocks", "em_restaged_waits"};
  for (int i = 0; i < 4; ++i)
    if (strcmp(name, em_names[i]) == 0) {
      unsigned long long v = 0;
      if (ctx->d_em_counters) {
        PENGK_HIP(hipMemcpyAsync(&v, ctx->d_em_counters + i, sizeof v, hipMemcpyDeviceToHost, ctx->stream));
        PENGK_HIP(hipStreamSynchronize(ctx->stream));
      }
      *value = (int64_t)v;
      return PENGK_OK;
    }
  return fail(PENGK_ERR_ARG, "unknown info '%s'", name);
}

void* pengk_stream(pengk_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int pengk_set_stream(pengk_ctx* ctx, void* s) {
  if (!ctx) return fail(PENGK_ERR_ARG, "ctx is NULL");
  PENGK_ENTER(ctx);
  PENGK_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->own_stream && ctx->stream) PENGK_HIP(hipStreamDestroy(ctx->stream));
  ctx->stream = (hipStream_t)s;
  ctx->own_stream = false;
  return PENGK_OK;
}

int pengk_malloc(pengk_ctx* ctx, size_t bytes, void** d_out) {
  if (!ctx || !d_out) return fail(PENGK_ERR_ARG, "pengk_malloc: NULL argument");
  PENGK_HIP(hipSetDevice(ctx->device));
  hipError_t e = hipMalloc(d_out, bytes ? bytes : 1);
  if (e != hipSuccess) return fail(PENGK_ERR_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
  return PENGK_OK;
}

int pengk_free(pengk_ctx* ctx, void* d_ptr) {
  if (!ctx) return fail(PENGK_ERR_ARG, "ctx is NULL");
  PENGK_ENTER(ctx);
  if (d_ptr) {
    PENGK_HIP(hipStreamSynchronize(ctx->stream));
    PENGK_HIP(hipFree(d_ptr));
  }
  return PENGK_OK;
}

int pengk_host_alloc(pengk_ctx* ctx, size_t bytes, void** h_out) {
  if (!ctx || !h_out) return fail(PENGK_ERR_ARG, "pengk_host_alloc: NULL argument");
  PENGK_HIP(hipSetDevice(ctx->device));
  hipError_t e = hipHostMalloc(h_out, bytes ? bytes : 1, hipHostMallocDefault);
  if (e != hipSuccess) return fail(PENGK_ERR_NOMEM, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
  return PENGK_OK;
}

int pengk_host_free(pengk_ctx* ctx, void* h_ptr) {
  if (!ctx) return fail(PENGK_ERR_ARG, "ctx is NULL");
  PENGK_ENTER(ctx);
  if (h_ptr) {
    PENGK_HIP(hipStreamSynchronize(ctx->stream));
    PENGK_HIP(hipHostFree(h_ptr));
  }
  return PENGK_OK;
}

int pengk_memcpy_h2d(pengk_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
  if (!ctx || (bytes && (!d_dst || !h_src))) return fail(PENGK_ERR_ARG, "pengk_memcpy_h2d: NULL argument");
  PENGK_ENTER(ctx);
  PENGK_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
  PENGK_HIP(hipStreamSynchronize(ctx->stream));
  return PENGK_OK;
}

// First uses cost: the first device-to-host copy of a process takes 10-17 ms whatever its size (the copy path of that
// direction is set up then), and every translation unit's code object is loaded at the first launch of one of its
// kernels.  A host that has something else to do meanwhile (the CLI: the upload of the packed sequences) calls this
// once, from any thread, right after pengk_create; nothing depends on it.
int pengk_warmup(pengk_ctx* ctx) {
  if (!ctx) return fail(PENGK_ERR_ARG, "ctx is NULL");
  PENGK_ENTER(ctx);
  // (on the context's own stream: a new stream is a new hardware queue, 20-40 ms on this host -- which is also why a
  // large upload split over several threads and streams, 22 instead of 35 ms in tools/ubench/runtime_start2.cpp, LOST in
  // the CLI: 120 ms instead of 18, profiles/r04_e2e_experiments.log)
  // (a copy of the size of a result table into page-locked memory: a few bytes do not take the path the tables take)
  constexpr size_t N = (size_t)4 << 20;
  void* d = nullptr;
  void* h = nullptr;
  hipError_t e = hipMalloc(&d, N);
  if (e == hipSuccess) e = hipHostMalloc(&h, N, hipHostMallocDefault);
  if (e == hipSuccess) e = hipMemcpyAsync(h, d, N, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (h) (void)hipHostFree(h);
  if (d) (void)hipFree(d);
  if (e != hipSuccess) return hip_fail(e, "pengk_warmup");
  return warm_code_objects();
}

int pengk_memcpy_d2h(pengk_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
  if (!ctx || (bytes && (!h_dst || !d_src))) return fail(PENGK_ERR_ARG, "pengk_memcpy_d2h: NULL argument");
  PENGK_ENTER(ctx);
  PENGK_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  PENGK_HIP(hipStreamSynchronize(ctx->stream));
  return PENGK_OK;
}

int pengk_memset(pengk_ctx* ctx, void* d_dst, int byte, size_t bytes) {
  if (!ctx || (bytes && !d_dst)) return fail(PENGK_ERR_ARG, "pengk_memset: NULL argument");
  PENGK_ENTER(ctx);
  PENGK_HIP(hipMemsetAsync(d_dst, byte, bytes, ctx->stream));
  return PENGK_OK;
}

int pengk_timer_create(pengk_ctx* ctx, void** out) {
  if (!ctx || !out) return fail(PENGK_ERR_ARG, "pengk_timer_create: NULL argument");
  PENGK_ENTER(ctx);
  hipEvent_t ev;
  PENGK_HIP(hipEventCreate(&ev));
  *out = (void*)ev;
  return PENGK_OK;
}
int pengk_timer_record(pengk_ctx* ctx, void* t) {
  if (!ctx || !t) return fail(PENGK_ERR_ARG, "pengk_timer_record: NULL argument");
  PENGK_ENTER(ctx);
  PENGK_HIP(hipEventRecord((hipEvent_t)t, ctx->stream));
  return PENGK_OK;
}
int pengk_timer_elapsed_ms(pengk_ctx* ctx, void* a, void* b, float* ms) {
  if (!ctx || !a || !b || !ms) return fail(PENGK_ERR_ARG, "pengk_timer_elapsed_ms: NULL argument");
  PENGK_ENTER(ctx);
  PENGK_HIP(hipEventSynchronize((hipEvent_t)b));
  PENGK_HIP(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
  return PENGK_OK;
}
int pengk_timer_destroy(pengk_ctx* ctx, void* t) {
  (void)ctx;
  if (t) PENGK_HIP(hipEventDestroy((hipEvent_t)t));
  return PENGK_OK;
}

// ---------------------------------------------------------------------------------------------
int pengk_set_sequences(pengk_ctx* ctx, const uint64_t* d_words, uint64_t n_words, const uint64_t* d_items,
                        uint64_t n_items, int W, int item_windows, uint64_t max_bin_bound, int all_whole) {
  if (!ctx) return fail(PENGK_ERR_ARG, "ctx is NULL");
  if (!valid_w(W)) return fail(PENGK_ERR_ARG, "pattern length %d unsupported (even, %d..%d)", W, PENGK_MIN_W, PENGK_MAX_W);
  if (item_windows < PENGK_MIN_ITEM_WINDOWS || item_windows > 65535)
    return fail(PENGK_ERR_ARG, "item_windows %d out of range [%d,65535]", item_windows, PENGK_MIN_ITEM_WINDOWS);
  if (!d_words || n_words < (PENGK_FRONT_PAD_BASES / 32) + 2) return fail(PENGK_ERR_ARG, "packed stream missing or too short");
  if (n_items && !d_items) return fail(PENGK_ERR_ARG, "items missing");
  if (n_items >= (1ull << 32)) return fail(PENGK_ERR_RANGE, "too many scan items for one shard (%llu)", (unsigned long long)n_items);
  ctx->d_words = d_words;
  ctx->n_words = n_words;
  ctx->d_items = d_items;
  ctx->n_items = n_items;
  ctx->W = W;
  ctx->item_windows = item_windows;
  ctx->max_bin_bound = max_bin_bound;
  ctx->all_whole = all_whole;
  ctx->n_windows_hint = 0;
  return PENGK_OK;
}

int pengk_synth_sizes(uint64_t n_seq, uint32_t L, int W, int item_windows, uint64_t* n_words, uint64_t* n_items) {
  if (!valid_w(W) || !n_words || !n_items) return fail(PENGK_ERR_ARG, "pengk_synth_sizes: bad argument");
  if (item_windows == 0) item_windows = PENGK_DEFAULT_ITEM_WINDOWS;
  if (L < (uint32_t)W) return fail(PENGK_ERR_ARG, "synthetic sequence length %u shorter than W=%d", L, W);
  const uint64_t nwin = (uint64_t)L - W + 1;
  const uint64_t per = (nwin + item_windows - 1) / item_windows;
  *n_items = n_seq * per;
  const uint64_t bases = PENGK_FRONT_PAD_BASES + n_seq * (uint64_t)L;
  *n_words = (bases + 31) / 32 + 4;
  return PENGK_OK;
}

int pengk_synth_sequences(pengk_ctx* ctx, uint64_t seed, uint64_t seq0, uint64_t n_seq, uint32_t L, int W,
                          int item_windows, uint64_t* d_words, uint64_t* d_items) {
  if (!ctx || !d_words || !d_items) return fail(PENGK_ERR_ARG, "pengk_synth_sequences: NULL argument");
  PENGK_ENTER(ctx);
  if (item_windows == 0) item_windows = PENGK_DEFAULT_ITEM_WINDOWS;
  uint64_t nw = 0, ni = 0;
  int rc = pengk_synth_sizes(n_seq, L, W, item_windows, &nw, &ni);
  if (rc) return rc;
  if (item_windows < PENGK_MIN_ITEM_WINDOWS || item_windows > 65535) return fail(PENGK_ERR_ARG, "item_windows out of range");
  rc = launch_synth(ctx, seed, seq0, n_seq, L, W, item_windows, d_words, d_items);
  if (rc) return rc;
  const uint64_t nwin = (uint64_t)L - W + 1;
  rc = pengk_set_sequences(ctx, d_words, nw, d_items, ni, W, item_windows, n_seq * ((nwin + W - 1) / W), 1);
  if (rc) return rc;
  ctx->n_windows_hint = n_seq * nwin;
  return PENGK_OK;
}

int pengk_count(pengk_ctx* ctx, int both, uint32_t* d_counts, uint64_t* d_ltot) {
  if (!ctx || !d_counts || !d_ltot) return fail(PENGK_ERR_ARG, "pengk_count: NULL argument");
  if (!ctx->d_words) return fail(PENGK_ERR_ARG, "pengk_count: no sequences attached");
  PENGK_ENTER(ctx);
  if (ctx->max_bin_bound >= (1ull << 32))
    return fail(PENGK_ERR_RANGE, "a count bin could reach %llu >= 2^32 on this shard; split the input",
                (unsigned long long)ctx->max_bin_bound);
  return launch_count(ctx, both ? 1 : 0, d_counts, d_ltot, nullptr);
}

int pengk_count_bg(pengk_ctx* ctx, int both, uint32_t* d_counts, uint64_t* d_ltot, uint64_t* d_bg) {
  if (!ctx || !d_counts || !d_ltot || !d_bg) return fail(PENGK_ERR_ARG, "pengk_count_bg: NULL argument");
  if (!ctx->d_words) return fail(PENGK_ERR_ARG, "pengk_count_bg: no sequences attached");
  PENGK_ENTER(ctx);
  if (!ctx->all_whole)
    return fail(PENGK_ERR_UNSUPPORTED, "pengk_count_bg: input has invalid bases or sequences shorter than W; use pengk_packed.bg_counts");
  if (ctx->max_bin_bound >= (1ull << 32))
    return fail(PENGK_ERR_RANGE, "a count bin could reach %llu >= 2^32 on this shard; split the input",
                (unsigned long long)ctx->max_bin_bound);
  return launch_count(ctx, both ? 1 : 0, d_counts, d_ltot, d_bg);
}

int pengk_mirror_counts(pengk_ctx* ctx, int W, uint32_t* d_counts) {
  if (!ctx || !d_counts) return fail(PENGK_ERR_ARG, "pengk_mirror_counts: NULL argument");
  PENGK_ENTER(ctx);
  if (!valid_w(W)) return fail(PENGK_ERR_ARG, "pattern length %d unsupported", W);
  return launch_mirror(ctx, W, d_counts);
}

int pengk_bg_count(pengk_ctx* ctx, uint64_t* d_bg) {
  if (!ctx || !d_bg) return fail(PENGK_ERR_ARG, "pengk_bg_count: NULL argument");
  if (!ctx->d_words) return fail(PENGK_ERR_ARG, "pengk_bg_count: no sequences attached");
  PENGK_ENTER(ctx);
  if (!ctx->all_whole)
    return fail(PENGK_ERR_UNSUPPORTED, "pengk_bg_count: input has invalid bases or sequences shorter than W; use pengk_packed.bg_counts");
  return launch_bg_count(ctx, d_bg);
}

int pengk_bg_model(pengk_ctx* ctx, const uint64_t* d_bg, int K, const float* h_alpha, float* d_V) {
  if (!ctx || !d_bg || !h_alpha || !d_V) return fail(PENGK_ERR_ARG, "pengk_bg_model: NULL argument");
  PENGK_ENTER(ctx);
  if (K < 0 || K > 2) return fail(PENGK_ERR_ARG, "background order %d unsupported (0..2)", K);
  return launch_bg_model(ctx, d_bg, K, h_alpha, d_V);
}

int pengk_pattern_stats(pengk_ctx* ctx, int W, int both, int k, int max_k, const float* d_V, const uint64_t* d_ltot,
                        const uint32_t* d_counts, float* d_bgprob, float* d_expected, float* d_logp, float* d_z) {
  if (!ctx || !d_V || !d_ltot || !d_counts || !d_bgprob || !d_expected || !d_logp || !d_z)
    return fail(PENGK_ERR_ARG, "pengk_pattern_stats: NULL argument");
  PENGK_ENTER(ctx);
  if (!valid_w(W)) return fail(PENGK_ERR_ARG, "pattern length %d unsupported", W);
  if (k < 0 || max_k > 2 || k > max_k || max_k > W - 1) return fail(PENGK_ERR_ARG, "background orders k=%d max_k=%d unsupported", k, max_k);
  return launch_stats(ctx, W, both ? 1 : 0, k, max_k, d_V, d_ltot, d_counts, d_bgprob, d_expected, d_logp, d_z);
}

int pengk_seed_candidates(pengk_ctx* ctx, int W, const float* d_z, const uint32_t* d_counts, float z_threshold,
                          uint64_t count_threshold, uint32_t* h_ids, float* h_z, int64_t capacity, int64_t* n_out) {
  if (!ctx || !d_z || !d_counts || !n_out || (capacity && (!h_ids || !h_z))) return fail(PENGK_ERR_ARG, "pengk_seed_candidates: NULL argument");
  if (!valid_w(W)) return fail(PENGK_ERR_ARG, "pattern length %d unsupported", W);
  if (capacity < 0 || capacity > (int64_t)1 << 28) return fail(PENGK_ERR_ARG, "pengk_seed_candidates: capacity out of range");
  PENGK_ENTER(ctx);
  const uint32_t cap = (uint32_t)capacity;
  const uint32_t cthr = count_threshold > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)count_threshold;
  const size_t need = 256 + (size_t)cap * 8;
  int rc = ensure_scratch(ctx, &ctx->d_misc, &ctx->misc_bytes, need);
  if (rc) return rc;
  uint32_t* d_n = (uint32_t*)ctx->d_misc;
  uint32_t* d_ids = (uint32_t*)((char*)ctx->d_misc + 256);
  float* d_zs = (float*)(d_ids + cap);
  rc = launch_seed_candidates(ctx, W, d_z, d_counts, z_threshold, cthr, cap, d_n, d_ids, d_zs);
  if (rc) return rc;
  uint32_t n = 0;
  PENGK_HIP(hipMemcpyAsync(&n, d_n, sizeof n, hipMemcpyDeviceToHost, ctx->stream));
  PENGK_HIP(hipStreamSynchronize(ctx->stream));
  *n_out = n;
  const uint32_t got = n < cap ? n : cap;  // n > capacity: the caller retries with room for n
  if (got) {
    PENGK_HIP(hipMemcpyAsync(h_ids, d_ids, (size_t)got * 4, hipMemcpyDeviceToHost, ctx->stream));
    PENGK_HIP(hipMemcpyAsync(h_z, d_zs, (size_t)got * 4, hipMemcpyDeviceToHost, ctx->stream));
    PENGK_HIP(hipStreamSynchronize(ctx->stream));
  }
  return PENGK_OK;
}

int pengk_iupac_aggregate(pengk_ctx* ctx, int W, int both, const uint64_t* h_ids, int64_t n, const uint32_t* d_counts,
                          const float* d_bgp, const float* d_expected, pengk_iupac_stats* h_out) {
  if (!ctx || (n && (!h_ids || !h_out)) || !d_counts || !d_bgp || !d_expected)
    return fail(PENGK_ERR_ARG, "pengk_iupac_aggregate: NULL argument");
  PENGK_ENTER(ctx);
  if (!valid_w(W)) return fail(PENGK_ERR_ARG, "pattern length %d unsupported", W);
  if (n < 0) return fail(PENGK_ERR_ARG, "negative pattern count");
  if (n == 0) return PENGK_OK;
  return launch_iupac(ctx, W, both ? 1 : 0, h_ids, n, d_counts, d_bgp, d_expected, h_out);
}

int pengk_motif_similarity(pengk_ctx* ctx, int n, const float* h_pwm, const float* h_comp, const int32_t* h_len,
                           const uint64_t* h_sites, int both, const float* h_bg, int first_new, float* h_out) {
  if (!ctx || !h_pwm || !h_comp || !h_len || !h_sites || !h_bg || !h_out) return fail(PENGK_ERR_ARG, "pengk_motif_similarity: NULL argument");
  if (n < 0 || first_new < 0 || first_new > n) return fail(PENGK_ERR_ARG, "pengk_motif_similarity: n = %d, first_new = %d", n, first_new);
  for (int i = 0; i < n; ++i)
    if (h_len[i] < 1 || h_len[i] > PENGK_MAX_MOTIF_LEN)
      return fail(PENGK_ERR_UNSUPPORTED, "pengk_motif_similarity: motif %d has %d columns (1..%d)", i, h_len[i], PENGK_MAX_MOTIF_LEN);
  long long pairs = 0;
  for (int j = first_new; j < n; ++j) pairs += j;
  if (pairs == 0) return PENGK_OK;
  PENGK_ENTER(ctx);
  const size_t pw = (size_t)n * PENGK_MAX_MOTIF_LEN * 4 * sizeof(float);
  const size_t o_len = 2 * pw, o_sites = (o_len + (size_t)n * 4 + 255) & ~(size_t)255, o_out = (o_sites + (size_t)n * 8 + 255) & ~(size_t)255;
  int rc = ensure_scratch(ctx, &ctx->d_sim, &ctx->sim_bytes, o_out + (size_t)pairs * sizeof(float));
  if (rc) return rc;
  char* base = (char*)ctx->d_sim;
  PENGK_HIP(hipMemcpyAsync(base, h_pwm, pw, hipMemcpyHostToDevice, ctx->stream));
  PENGK_HIP(hipMemcpyAsync(base + pw, h_comp, pw, hipMemcpyHostToDevice, ctx->stream));
  PENGK_HIP(hipMemcpyAsync(base + o_len, h_len, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  PENGK_HIP(hipMemcpyAsync(base + o_sites, h_sites, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
  rc = launch_similarity(ctx, n, (const float*)base, (const float*)(base + pw), (const int32_t*)(base + o_len),
                         (const uint64_t*)(base + o_sites), both ? 1 : 0, h_bg, first_new, (float*)(base + o_out));
  if (rc) return rc;
  PENGK_HIP(hipMemcpyAsync(h_out, base + o_out, (size_t)pairs * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  PENGK_HIP(hipStreamSynchronize(ctx->stream));
  return PENGK_OK;
}

int pengk_sequential_sum_f32(pengk_ctx* ctx, const float* d_terms, uint64_t n_chains, uint64_t chain_len, float* d_out) {
  if (!ctx || (!d_terms && n_chains && chain_len) || (!d_out && n_chains)) return fail(PENGK_ERR_ARG, "pengk_sequential_sum_f32: NULL argument");
  PENGK_ENTER(ctx);
  return launch_sequential_sum(ctx, d_terms, n_chains, chain_len, d_out);
}

int pengk_test_em_generation(pengk_ctx* ctx, int generation) {
  if (!ctx) return fail(PENGK_ERR_ARG, "pengk_test_em_generation: NULL context");
  if (generation < 0 || generation > 3) return fail(PENGK_ERR_ARG, "pengk_test_em_generation: 0 (dependent additions), 1 (scan), 2 / 3 (the library's scheme)");
  ctx->em_serial_scan = generation;
  return PENGK_OK;
}

int pengk_em_device(pengk_ctx* ctx, int W, int64_t n_pwm, float* d_pwms, float saturation, float threshold, int max_it,
                    const uint32_t* d_counts, const float* d_bg, int32_t* d_state, float* d_change) {
  if (!ctx || !d_pwms || !d_counts || !d_bg || !d_state || !d_change) return fail(PENGK_ERR_ARG, "pengk_em_device: NULL argument");
  PENGK_ENTER(ctx);
  if (!valid_w(W)) return fail(PENGK_ERR_ARG, "pattern length %d unsupported", W);
  if (n_pwm < 0) return fail(PENGK_ERR_ARG, "negative count");
  if (max_it < 0) max_it = 0;  // (the reference's loop ends at `iteration_counter >= max_iterations`, src/peng.cpp:104: no iteration)
  if (n_pwm == 0) return PENGK_OK;
  return launch_em(ctx, W, n_pwm, d_pwms, saturation, threshold, max_it, d_counts, d_bg, d_state, d_change);
}

int pengk_em(pengk_ctx* ctx, int W, int64_t n_pwm, float* h_pwms, float saturation, float threshold, int max_it,
             const uint32_t* d_counts, const float* d_bg, int* h_iters, float* h_change) {
  if (!ctx || (n_pwm && !h_pwms) || !d_counts || !d_bg) return fail(PENGK_ERR_ARG, "pengk_em: NULL argument");
  PENGK_ENTER(ctx);
  if (!valid_w(W)) return fail(PENGK_ERR_ARG, "pattern length %d unsupported", W);
  if (n_pwm < 0) return fail(PENGK_ERR_ARG, "negative count");
  if (max_it < 0) max_it = 0;  // (the reference's loop ends at `iteration_counter >= max_iterations`, src/peng.cpp:104: no iteration)
  if (n_pwm == 0) return PENGK_OK;
  const size_t pw = (size_t)n_pwm * W * 4 * sizeof(float);
  const size_t st = (size_t)n_pwm * 2 * sizeof(int32_t);
  const size_t ch = (size_t)n_pwm * sizeof(float);
  const size_t need = ((pw + 255) & ~(size_t)255) + ((st + 255) & ~(size_t)255) + ch;
  int rc = ensure_scratch(ctx, &ctx->d_misc, &ctx->misc_bytes, need);
  if (rc) return rc;
  float* d_pwms = (float*)ctx->d_misc;
  int32_t* d_state = (int32_t*)((char*)ctx->d_misc + ((pw + 255) & ~(size_t)255));
  float* d_change = (float*)((char*)d_state + ((st + 255) & ~(size_t)255));
  PENGK_HIP(hipMemcpyAsync(d_pwms, h_pwms, pw, hipMemcpyHostToDevice, ctx->stream));
  rc = launch_em(ctx, W, n_pwm, d_pwms, saturation, threshold, max_it, d_counts, d_bg, d_state, d_change);
  if (rc) return rc;
  PENGK_HIP(hipMemcpyAsync(h_pwms, d_pwms, pw, hipMemcpyDeviceToHost, ctx->stream));
  PENGK_HIP(hipStreamSynchronize(ctx->stream));
  if (h_iters || h_change) {
    int32_t* tmp = new (std::nothrow) int32_t[(size_t)n_pwm * 2];
    if (!tmp) return fail(PENGK_ERR_NOMEM, "pengk_em: out of host memory");
    hipError_t e = hipMemcpy(tmp, d_state, st, hipMemcpyDeviceToHost);
    if (e == hipSuccess && h_change) e = hipMemcpy(h_change, d_change, ch, hipMemcpyDeviceToHost);
    if (h_iters)
      for (int64_t i = 0; i < n_pwm; ++i) h_iters[i] = tmp[2 * i];
    delete[] tmp;
    if (e != hipSuccess) return hip_fail(e, "pengk_em: copy back");
  }
  return PENGK_OK;
}

}  // extern "C"
