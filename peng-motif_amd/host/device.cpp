#include "device.h"

#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <deque>
#include <mutex>
#include <cstdlib>
#include <iostream>
#include <string>
#include <thread>

#include "Global.h"
#include "shared/SequenceSet.h"

namespace pengk_host {

static pengk_ctx* g_ctx = nullptr;

static int env_int(const char* name, int fallback) {
  const char* e = std::getenv(name);
  return e && *e ? std::atoi(e) : fallback;
}
int world() {
  static const int w = std::max(1, env_int("WORLD_SIZE", 1));
  return w;
}
bool launched() {
  static const bool l = std::getenv("WORLD_SIZE") != nullptr;
  return l;
}
int rank() {
  static const int r = launched() ? env_int("RANK", 0) : 0;
  return r;
}
static int device_index() { return launched() && std::getenv("LOCAL_RANK") ? env_int("LOCAL_RANK", 0) : Global::device; }

static bool shard_allgather(const void* send, void* recv, size_t bytes) {
  return pengk_comm_host_allgather(send, recv, bytes) == PENGK_OK;
}

void start_sharded_ingest() {
  if (world() <= 1) return;
  check(pengk_comm_host_init_env(), "pengk_comm_host_init_env");
  SequenceShardComm sc;
  sc.rank = rank();
  sc.world = world();
  sc.allgather = shard_allgather;
  SequenceSet::setShardComm(sc);
}

void finish_ranks() {
  if (!launched()) return;
  uint64_t one = 1;
  if (world() > 1) check(pengk_comm_host_allreduce_u64(&one, 1), "pengk_comm_host_allreduce_u64");  // nobody leaves early
  if (g_ctx) check(pengk_comm_destroy(g_ctx), "pengk_comm_destroy");
  pengk_comm_host_shutdown();
}

void check(int rc, const char* what) {
  if (rc == PENGK_OK) return;
  std::cerr << "Error: " << what << " failed: " << pengk_error_name(rc) << ": " << pengk_last_error() << std::endl;
  exit(1);
}

// Context creation (HIP runtime start-up + code object load, ~0.2 s) can run beside the FASTA reader.
static std::thread g_starter;
static int g_starter_rc = PENGK_OK;

static void join_starter() {
  if (g_starter.joinable()) g_starter.join();
}

void start_context() {
  if (g_ctx || g_starter.joinable()) return;
  g_starter = std::thread([] {
    g_starter_rc = pengk_create(device_index(), &g_ctx);
    if (g_starter_rc != PENGK_OK) {
      // no usable gfx950 device: there is no CPU path, and no point in reading gigabytes of FASTA first
      std::cerr << "Error: pengk_create failed: " << pengk_error_name(g_starter_rc) << ": " << pengk_last_error() << std::endl;
      _exit(1);
    }
  });
  atexit(join_starter);  // an exit() on a FASTA error must not tear the process down under a starting runtime
}

pengk_ctx* context() {
  if (g_starter.joinable()) {
    g_starter.join();  // (a failed start has already ended the process)
  }
  if (!g_ctx) check(pengk_create(device_index(), &g_ctx), "pengk_create");
  static bool options_set = false;
  if (!options_set) {
    options_set = true;
    // The CLI runs the EM in the library's serial mode (em_fast = 2): the reference's float32 arithmetic including its
    // summation order, so that the merge / redundancy decisions downstream -- exact ties in real arithmetic for
    // reverse-complement twins -- fall as the reference's do.  PENGK_EM_FAST=1 (one reciprocal per weight, fp64 tree
    // sums, ~10x faster for large PWM sets) or 0 (reference terms, fp64 tree sums) select the throughput modes.
    int mode = 2;
    if (const char* e = std::getenv("PENGK_EM_FAST")) mode = std::atoi(e);
    check(pengk_set_option(g_ctx, "em_fast", mode), "pengk_set_option");
    if (launched()) {  // RCCL over xGMI; collective: every rank gets here
      // librccl announces its version on stdout while the communicator is built; stdout is the reference's trace
      // (compared byte for byte), so that line goes to stderr
      std::cout.flush();
      fflush(stdout);
      const int saved = dup(1);
      if (saved >= 0) dup2(2, 1);
      const int rc = pengk_comm_init_env(g_ctx);
      fflush(stdout);
      if (saved >= 0) {
        dup2(saved, 1);
        close(saved);
      }
      check(rc, "pengk_comm_init_env");
    }
  }
  return g_ctx;
}

// ---- streaming pack ------------------------------------------------------------------------------------------------
namespace {
struct Stream {
  PackedInput in;
  SequenceSet* set = nullptr;
  size_t words_cap = 0, items_cap = 0;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<pengk_packed> ready;  // packed chunks waiting for the uploader
  bool done = false;
  std::string error;               // first packer / upload failure
  std::thread uploader;
  bool started = false;
};
Stream* g_stream = nullptr;

void stream_fail(Stream* st, const std::string& what) {
  std::lock_guard<std::mutex> lock(st->mu);
  if (st->error.empty()) st->error = what;
}

void sink_begin(void* user, size_t range_bytes, size_t n_chunks) {
  Stream* st = (Stream*)user;
  // upper bounds from the bytes of text: a base takes a byte; an item needs a run of >= W bases plus a byte that ends
  // it, and covers up to item_windows windows
  const size_t W = (size_t)st->in.W, M = (size_t)PENGK_DEFAULT_ITEM_WINDOWS;
  st->words_cap = range_bytes / 32 + n_chunks * 8 + 16;
  st->items_cap = range_bytes / (W + 1) + range_bytes / M + n_chunks + 16;
  st->started = true;
  st->uploader = std::thread([st] {
    uint64_t word_at = 0, item_at = 0;
    for (;;) {
      pengk_packed pk;
      {
        std::unique_lock<std::mutex> lock(st->mu);
        st->cv.wait(lock, [st] { return !st->ready.empty() || st->done; });
        if (st->ready.empty()) return;
        pk = st->ready.front();
        st->ready.pop_front();
      }
      bool ok;
      {
        std::lock_guard<std::mutex> lock(st->mu);
        ok = st->error.empty();
      }
      if (ok && !st->in.d_words) {  // first chunk: the context (started beside the reader) is needed from here on
        void *w = nullptr, *it = nullptr;
        ok = pengk_malloc(context(), st->words_cap * sizeof(uint64_t), &w) == PENGK_OK &&
             pengk_malloc(context(), st->items_cap * sizeof(uint64_t), &it) == PENGK_OK;
        st->in.d_words = (uint64_t*)w;
        st->in.d_items = (uint64_t*)it;
      }
      if (ok && (word_at + pk.n_words > st->words_cap || item_at + pk.n_items > st->items_cap)) {
        stream_fail(st, "packed chunks exceed the bounds computed from the file size");
        ok = false;
      }
      if (ok && pengk_append_packed(context(), st->in.d_words, word_at, st->in.d_items, item_at, &pk) != PENGK_OK) ok = false;
      if (!ok) stream_fail(st, std::string("upload of a packed chunk failed: ") + pengk_last_error());
      word_at += pk.n_words;
      item_at += pk.n_items;
      st->in.n_words = word_at;
      st->in.n_items = item_at;
      pengk_packed_free(&pk);
    }
  });
}

void sink_chunk(void* user, size_t, const SequenceChunk& c) {
  Stream* st = (Stream*)user;
  if (c.n == 0) return;
  pengk_packed pk;
  if (pengk_pack_threads(c.codes, c.offs.data(), (int64_t)c.n, st->in.W, 0, 1, &pk) != PENGK_OK) {
    stream_fail(st, std::string("pengk_pack failed: ") + pengk_last_error());
    return;
  }
  {
    std::lock_guard<std::mutex> lock(st->mu);
    st->in.item_windows = pk.item_windows;
    st->in.n_windows += pk.n_windows;
    st->in.max_bin_bound += pk.max_bin_bound;
    st->in.all_whole &= pk.all_whole;
    for (int i = 0; i < 84; ++i) st->in.bg_counts[i] += pk.bg_counts[i];
    st->ready.push_back(pk);
  }
  st->cv.notify_one();
}
}  // namespace

void begin_streaming_pack(int W) {
  if (const char* e = std::getenv("PENGK_NO_STREAMING")) {  // the staged path: read, then pack, then upload (tests compare the two)
    if (std::atoi(e)) return;
  }
  delete g_stream;
  g_stream = new Stream();
  g_stream->in.W = W;
  SequenceChunkSink sink;
  sink.user = g_stream;
  sink.begin = sink_begin;
  sink.chunk = sink_chunk;
  SequenceSet::setChunkSink(sink);
}

const PackedInput* finish_streaming_pack(SequenceSet* set) {
  Stream* st = g_stream;
  if (!st || !st->started) return nullptr;
  {
    std::lock_guard<std::mutex> lock(st->mu);
    st->done = true;
  }
  st->cv.notify_all();
  st->uploader.join();
  if (!st->error.empty()) {
    std::cerr << "Error: " << st->error << std::endl;
    exit(1);
  }
  st->set = set;
  if (!st->in.d_words) return nullptr;  // no records at all on this rank: the staged path handles the empty shard
  return &st->in;
}

const PackedInput* packed_input(SequenceSet* set, int W) {
  Stream* st = g_stream;
  return st && st->set == set && st->in.W == W && st->in.d_words ? &st->in : nullptr;
}

void shutdown() {
  if (g_stream && g_ctx) {
    if (g_stream->in.d_words) pengk_free(g_ctx, g_stream->in.d_words);
    if (g_stream->in.d_items) pengk_free(g_ctx, g_stream->in.d_items);
  }
  delete g_stream;
  g_stream = nullptr;
  if (g_ctx) pengk_destroy(g_ctx);
  g_ctx = nullptr;
}

}  // namespace pengk_host
