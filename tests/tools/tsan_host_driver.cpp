// ThreadSanitizer harness for the THREADED host code of the path (build container only; CPU build, no GPU):
//
//   tsan_host_driver ingest FASTA W
//       the CLI's streaming ingest as it ships -- host/device.cpp (context starter thread, uploader thread, the sink the
//       reader's threads call), host/shared/SequenceSet.cpp (chunked reader, madvise of packed chunks) and
//       csrc/pack.cpp (pengk_pack_append: lock-free cursors into one pair of buffers) -- linked against the stand-in
//       "device" below (pengk_malloc = malloc, pengk_memcpy_h2d = memcpy, pengk_create = a context that takes a moment to
//       come up), and compared with the staged pack of the same file: same words, same items (as sets), same counters.
//   tsan_host_driver chan
//       the host channel of csrc/comm.hip under RANK / WORLD_SIZE / MASTER_* from the environment: four threads per
//       rank call the collectives at once (the library serialises them; what pairs with what across ranks is arbitrary,
//       so the threaded calls are all of one kind and size, every one contributes the same value and every result is checked).
//
// tests/tools/tsan_host.sh builds and runs both with forced chunkings and 1 / 2 / 4 ranks.
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "pengk_internal.h"
// (host mirror headers)
#include "Global.h"
#include "device.h"
#include "shared/SequenceSet.h"

// ---- the stand-in device: the few entry points of csrc/api.hip the ingest path calls -----------------------------------
namespace pengk {
static thread_local char g_err[512];
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}
int hip_fail(hipError_t, const char* what) { return fail(PENGK_ERR_DEVICE, "%s", what); }
int enter(pengk_ctx*) { return PENGK_OK; }
int ensure_scratch(pengk_ctx*, void**, size_t*, size_t) { return PENGK_ERR_DEVICE; }
}  // namespace pengk

extern "C" {
const char* pengk_last_error(void) { return pengk::g_err; }
const char* pengk_error_name(int) { return "PENGK_ERR"; }
int pengk_create(int, pengk_ctx** out) {
  usleep(30000);  // the runtime takes a while to come up: the reader and the packer run meanwhile
  *out = new pengk_ctx();
  return PENGK_OK;
}
int pengk_destroy(pengk_ctx* c) {
  delete c;
  return PENGK_OK;
}
int pengk_warmup(pengk_ctx*) { return PENGK_OK; }
int pengk_set_option(pengk_ctx*, const char*, int64_t) { return PENGK_OK; }
int pengk_synchronize(pengk_ctx*) { return PENGK_OK; }
int pengk_malloc(pengk_ctx*, size_t bytes, void** out) {
  *out = malloc(bytes ? bytes : 1);
  return *out ? PENGK_OK : PENGK_ERR_NOMEM;
}
int pengk_free(pengk_ctx*, void* p) {
  free(p);
  return PENGK_OK;
}
int pengk_memcpy_h2d(pengk_ctx*, void* d, const void* h, size_t n) {
  memcpy(d, h, n);
  return PENGK_OK;
}
int pengk_memcpy_d2h(pengk_ctx*, void* h, const void* d, size_t n) {
  memcpy(h, d, n);
  return PENGK_OK;
}
}

static int run_ingest(const char* fasta, int W) {
  Alphabet::init("STANDARD");
  // 1. the streaming path, exactly as Global::init drives it
  pengk_host::start_context();
  pengk_host::begin_streaming_pack(W);
  SequenceSet* streamed = new SequenceSet(fasta, true);
  const pengk_host::PackedInput* in = pengk_host::finish_streaming_pack(streamed);
  // 2. the staged path on a second read of the file (no sink)
  SequenceSet staged(fasta, true);
  pengk_packed pk;
  if (pengk_pack(staged.codes(), staged.offsets(), (int64_t)staged.getLocalN(), W, 0, &pk) != PENGK_OK) {
    fprintf(stderr, "pengk_pack failed: %s\n", pengk_last_error());
    return 1;
  }
  if (!in) {
    const bool empty = pk.n_items == 0;
    printf("ingest %s: nothing streamed (%s)\n", fasta, empty ? "empty input: ok" : "MISMATCH");
    return empty ? 0 : 1;
  }
  // chunks land in the order their threads finish: compare what does not depend on it -- the totals, the counters, and
  // the multiset of windows each item describes (the 2W-bit ids of its first window, read from its own stream)
  bool ok = in->n_items == pk.n_items && in->n_windows == pk.n_windows && in->max_bin_bound == pk.max_bin_bound &&
            in->all_whole == pk.all_whole;
  for (int i = 0; i < 84; ++i) ok &= in->bg_counts[i] == pk.bg_counts[i];
  auto signature = [&](const uint64_t* words, const uint64_t* items, uint64_t n) {
    std::vector<uint64_t> sig((size_t)n);
    for (uint64_t i = 0; i < n; ++i) {
      const uint64_t it = items[i], ws = it & pengk::ITEM_WS_MASK, nw = (it >> pengk::ITEM_NW_SHIFT) & pengk::ITEM_NW_MASK;
      uint64_t id = 0;  // first window's id (ws counts bases behind the front pad of the stream)
      for (int p = 0; p < W; ++p) {
        const uint64_t g = ws + (uint64_t)p;
        id |= ((words[g / 32] >> (2 * (g % 32))) & 3ull) << (2 * p);
      }
      sig[(size_t)i] = (id << 20) | (nw << 1) | ((it >> pengk::ITEM_CONT_SHIFT) & 1ull);
    }
    std::sort(sig.begin(), sig.end());
    return sig;
  };
  if (ok) ok = signature(in->d_words, in->d_items, in->n_items) == signature(pk.words, pk.items, pk.n_items);
  printf("ingest %s W=%d: %llu items, %llu windows: %s\n", fasta, W, (unsigned long long)pk.n_items, (unsigned long long)pk.n_windows,
         ok ? "streamed == staged" : "MISMATCH");
  pengk_packed_free(&pk);
  delete streamed;
  pengk_host::shutdown();
  return ok ? 0 : 1;
}

static int run_chan() {
  if (pengk_comm_host_init_env() != PENGK_OK) {
    fprintf(stderr, "host channel: %s\n", pengk_last_error());
    return 1;
  }
  int rank = 0, world = 1;
  pengk_comm_host_info(&rank, &world);
  std::atomic<int> bad{0};
  std::vector<std::thread> th;
  for (int t = 0; t < 4; ++t)
    th.emplace_back([&] {
      for (int i = 0; i < 25; ++i) {
        uint64_t v[3] = {1, 10, 1ull << 40};
        if (pengk_comm_host_allreduce_u64(v, 3) != PENGK_OK || v[0] != (uint64_t)world || v[1] != 10ull * world ||
            v[2] != ((uint64_t)world << 40))
          ++bad;
        int r2 = -1, w2 = -1;
        pengk_comm_host_info(&r2, &w2);
        if (r2 != rank || w2 != world) ++bad;
      }
    });
  for (auto& t : th) t.join();
  // (calls of different kinds pair up across ranks only when every rank issues them in one order: single-threaded)
  int32_t mine = 7 + rank;
  std::vector<int32_t> all((size_t)world, 0);
  if (pengk_comm_host_allgather(&mine, all.data(), sizeof mine) != PENGK_OK) ++bad;
  for (int r = 0; r < world; ++r) bad += all[(size_t)r] != 7 + r;
  uint64_t one = 1;
  if (pengk_comm_host_allreduce_u64(&one, 1) != PENGK_OK) ++bad;  // nobody leaves early
  pengk_comm_host_shutdown();
  printf("chan rank %d of %d: %s\n", rank, world, bad ? "MISMATCH" : "ok");
  return bad ? 1 : 0;
}

int main(int argc, char** argv) {
  if (argc >= 4 && !strcmp(argv[1], "ingest")) return run_ingest(argv[2], atoi(argv[3]));
  if (argc >= 2 && !strcmp(argv[1], "chan")) return run_chan();
  fprintf(stderr, "usage: tsan_host_driver ingest FASTA W | chan\n");
  return 2;
}
