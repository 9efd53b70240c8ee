#include "device.h"

#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <mutex>
#include <cstdlib>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

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
  // A communicator set-up whose deadline passed has left a helper thread inside ncclCommInitRank: no exit handlers, no
  // static destructors of HIP / RCCL under it (they can crash or hang, and a hang would defeat the deadline)
  if (pengk_comm_init_abandoned()) {
    std::cerr.flush();
    std::cout.flush();
    _exit(1);
  }
  exit(1);
}

// Context creation (HIP runtime start-up + code object load, ~0.2 s) can run beside the FASTA reader.
static std::thread g_starter, g_warmer;
static int g_starter_rc = PENGK_OK;

static void join_starter() {
  if (g_starter.joinable()) g_starter.join();
  if (g_warmer.joinable()) g_warmer.join();
}

// seconds since the process image started (PENGK_TIMING: when did the context come up, when was the upload done)
static double since_start() {
  static const auto t0 = std::chrono::steady_clock::now();
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
static const double g_clock_anchor = since_start();

void start_context() {
  if (g_ctx || g_starter.joinable()) return;
  g_starter = std::thread([] {
    g_starter_rc = pengk_create(device_index(), &g_ctx);
    if (g_starter_rc != PENGK_OK) {
      // no usable gfx950 device: there is no CPU path, and no point in reading gigabytes of FASTA first
      std::cerr << "Error: pengk_create failed: " << pengk_error_name(g_starter_rc) << ": " << pengk_last_error() << std::endl;
      _exit(1);
    }
    if (std::getenv("PENGK_TIMING")) std::cerr << "[timing] (device context up at " << since_start() << " s)" << std::endl;
    // first-use costs that nothing has to wait for (the first device-to-host copy of a process: 10-17 ms; the code objects
    // of the sweep / IUPAC / EM kernels) are paid beside the upload of the packed sequences
    if (!std::getenv("PENGK_NO_WARMUP")) g_warmer = std::thread([] { (void)pengk_warmup(g_ctx); });
  });
  atexit(join_starter);  // an exit() on a FASTA error must not tear the process down under a starting runtime
}

// the context itself, without the communicator of a multi-rank run: what the uploader thread of the streaming pack
// may call while the main thread is busy on the host channel
static std::mutex g_ctx_mu;
static pengk_ctx* raw_context() {
  std::lock_guard<std::mutex> lock(g_ctx_mu);
  if (g_starter.joinable()) g_starter.join();  // (a failed start has already ended the process)
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
  }
  return g_ctx;
}

pengk_ctx* context() {
  pengk_ctx* ctx = raw_context();
  static bool comm_set = false;  // main thread only: the communicator is built by a collective over the ranks
  if (!comm_set) {
    comm_set = true;
    if (launched()) {  // RCCL over xGMI; collective: every rank gets here
      // librccl announces its version on stdout while the communicator is built; stdout is the reference's trace
      // (compared byte for byte), so that line goes to stderr
      std::cout.flush();
      fflush(stdout);
      const int saved = dup(1);
      if (saved >= 0) dup2(2, 1);
      const int rc = pengk_comm_init_env(ctx);
      fflush(stdout);
      if (saved >= 0) {
        dup2(saved, 1);
        close(saved);
      }
      check(rc, "pengk_comm_init_env");
    }
  }
  return ctx;
}

// ---- streaming pack ------------------------------------------------------------------------------------------------
namespace {
constexpr uint64_t SLAB_WORDS = (uint64_t)4 << 20;  // the uploader sends the stream in pieces of 32 MiB as they fill up

struct Stream {
  PackedInput in;
  SequenceSet* set = nullptr;
  pengk_pack_target target{};      // host buffers that collect the chunks (fresh anonymous memory: zero-filled)
  size_t words_bytes = 0, items_bytes = 0;
  std::mutex mu;
  std::condition_variable cv;
  std::vector<uint64_t> slab_fill; // words written into every slab so far
  uint64_t reserved_end = 0;       // highest word any finished chunk reaches
  bool done = false;
  std::string error;               // first packer / upload failure
  std::thread uploader;
  bool started = false;
  ~Stream() {
    if (target.words) munmap(target.words, words_bytes);
    if (target.items) munmap(target.items, items_bytes);
  }
};
Stream* g_stream = nullptr;

void stream_fail(Stream* st, const std::string& what) {
  std::lock_guard<std::mutex> lock(st->mu);
  if (st->error.empty()) st->error = what;
}

void* map_zero(size_t bytes) {
  void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  if (p == MAP_FAILED) return nullptr;
  if (!std::getenv("PENGK_NO_HUGEPAGES")) madvise(p, bytes, MADV_HUGEPAGE);
  return p;
}

// slabs [from, ...) that are full -- or, once the reader is done, everything up to the end -- go to the device
void upload_loop(Stream* st) {
  uint64_t sent = 0;  // words [0, sent) are on the device
  for (;;) {
    uint64_t upto = sent;
    bool last = false;
    {
      std::unique_lock<std::mutex> lock(st->mu);
      for (;;) {
        upto = sent;
        while (upto / SLAB_WORDS < st->slab_fill.size() && st->slab_fill[upto / SLAB_WORDS] == SLAB_WORDS) upto += SLAB_WORDS;
        if (upto > sent || st->done) break;
        st->cv.wait(lock);
      }
      if (upto == sent) {  // done, and no further full slab: the rest in one piece
        upto = st->reserved_end;
        last = true;
      }
      if (!st->error.empty()) return;
    }
    if (upto > sent) {
      if (!st->in.d_words) {  // first piece: the context (started beside the reader) is needed from here on
        void* w = nullptr;
        if (pengk_malloc(raw_context(), st->target.words_cap * sizeof(uint64_t), &w) != PENGK_OK) {
          stream_fail(st, std::string("device buffer for the packed stream: ") + pengk_last_error());
          return;
        }
        st->in.d_words = (uint64_t*)w;
      }
      if (pengk_memcpy_h2d(raw_context(), st->in.d_words + sent, st->target.words + sent, (upto - sent) * sizeof(uint64_t)) != PENGK_OK) {
        stream_fail(st, std::string("upload of the packed stream failed: ") + pengk_last_error());
        return;
      }
      sent = upto;
    }
    if (last) break;
  }
  // the items: one piece, at the end (a sixth of the stream's bytes)
  const uint64_t n_items = __atomic_load_n(&st->target.item_cursor, __ATOMIC_RELAXED);
  if (st->in.d_words) {
    void* it = nullptr;
    if (pengk_malloc(raw_context(), (n_items + 1) * sizeof(uint64_t), &it) != PENGK_OK ||
        (n_items && pengk_memcpy_h2d(raw_context(), it, st->target.items, n_items * sizeof(uint64_t)) != PENGK_OK)) {
      stream_fail(st, std::string("upload of the scan items failed: ") + pengk_last_error());
      return;
    }
    st->in.d_items = (uint64_t*)it;
  }
  st->in.n_words = sent;
  st->in.n_items = n_items;
  if (std::getenv("PENGK_TIMING")) std::cerr << "[timing] (packed sequences on the device at " << since_start() << " s)" << std::endl;
}

void sink_begin(void* user, size_t range_bytes, size_t n_chunks) {
  Stream* st = (Stream*)user;
  // upper bounds from the bytes of text: a base takes a byte; an item needs a run of >= W bases plus a byte that ends
  // it, and covers up to item_windows windows.  Address space only: pages are touched as chunks land on them.
  const size_t W = (size_t)st->in.W, M = (size_t)PENGK_DEFAULT_ITEM_WINDOWS;
  st->target.words_cap = range_bytes / 32 + n_chunks * 8 + 16;
  st->target.items_cap = range_bytes / (W + 1) + range_bytes / M + n_chunks + 16;
  st->words_bytes = st->target.words_cap * sizeof(uint64_t);
  st->items_bytes = st->target.items_cap * sizeof(uint64_t);
  st->target.words = (uint64_t*)map_zero(st->words_bytes);
  st->target.items = (uint64_t*)map_zero(st->items_bytes);
  if (!st->target.words || !st->target.items) {
    stream_fail(st, "out of memory for the packed stream");
    return;
  }
  st->slab_fill.assign((size_t)(st->target.words_cap / SLAB_WORDS + 1), 0);
  st->started = true;
  st->uploader = std::thread(upload_loop, st);
}

// (returns true: the chunk's byte codes are packed, the reader may give them back)
bool sink_chunk(void* user, size_t, const SequenceChunk& c) {
  Stream* st = (Stream*)user;
  if (c.n == 0 || !st->started) return false;
  pengk_packed pk;
  if (pengk_pack_append(c.codes, c.offs.data(), (int64_t)c.n, st->in.W, 0, &st->target, &pk) != PENGK_OK) {
    stream_fail(st, std::string("pengk_pack_append failed: ") + pengk_last_error());
    return false;
  }
  bool filled = false;
  {
    std::lock_guard<std::mutex> lock(st->mu);
    st->in.item_windows = pk.item_windows;
    st->in.n_windows += pk.n_windows;
    st->in.max_bin_bound += pk.max_bin_bound;
    st->in.all_whole &= pk.all_whole;
    for (int i = 0; i < 84; ++i) st->in.bg_counts[i] += pk.bg_counts[i];
    uint64_t w = (uint64_t)(pk.words - st->target.words);
    const uint64_t end = w + pk.n_words;
    if (end > st->reserved_end) st->reserved_end = end;
    while (w < end) {  // the chunk's words, slab by slab
      const uint64_t slab = w / SLAB_WORDS, upto = std::min(end, (slab + 1) * SLAB_WORDS);
      st->slab_fill[slab] += upto - w;
      filled |= st->slab_fill[slab] == SLAB_WORDS;
      w = upto;
    }
  }
  if (filled) st->cv.notify_one();
  return true;
}
}  // namespace

void begin_streaming_pack(int W) {
  if (W < PENGK_MIN_W || W > PENGK_MAX_W) return;  // (Peng::process ends such a run with the reference's message)
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
  if (!st) return nullptr;
  if (std::getenv("PENGK_TIMING")) std::cerr << "[timing] (reader done at " << since_start() << " s)" << std::endl;
  if (st->started) {
    {
      std::lock_guard<std::mutex> lock(st->mu);
      st->done = true;
    }
    st->cv.notify_all();
    st->uploader.join();
  }
  if (!st->error.empty()) {
    std::cerr << "Error: " << st->error << std::endl;
    exit(1);
  }
  st->set = set;
  // The host copies have done their duty, but they are NOT handed back here: with the reader's threads just gone, an
  // munmap of 0.6 GB costs ~35 ms of TLB shoot-downs on the bench's input -- on this thread, or, from a helper thread, as
  // 33 ms of a blocked pengk_set_sequences -- while the exiting process returns the same pages for nothing
  // (tools/e2e_ab.sh, profiles/r04_e2e_experiments.log: median 0.45 / 0.43 / 0.32 s with sync / async / no release on a
  // noisy box).  ~Stream (PENGK_FULL_TEARDOWN) releases them.
  // That trade is for inputs of the bench's size.  A large input must not keep bases / 4 bytes plus the item table mapped
  // for the rest of the run (~5 GB for a single-process 100M x 200 bp input, ~0.6 GB per configs[3] shard): above
  // PENGK_HOST_RELEASE_MB (default 1024 MiB) the host copies go back now -- such a run takes seconds, the shoot-downs
  // ~60 ms per GB.  0 = never (the round-4 behaviour), 1 = always.
  {
    size_t limit_mb = 1024;
    if (const char* e = std::getenv("PENGK_HOST_RELEASE_MB")) limit_mb = (size_t)std::strtoull(e, nullptr, 10);
    // (what is RESIDENT: the mappings are reserved for the worst case the file size allows -- 2 GiB for the bench's 2.1 GB
    // file -- and only the words and items actually written were ever touched: 0.58 GB there)
    const size_t held = (size_t)(st->in.n_words + st->in.n_items) * sizeof(uint64_t);
    if (limit_mb != 0 && (limit_mb == 1 || held > (limit_mb << 20))) {
      if (st->target.words) munmap(st->target.words, st->words_bytes);
      if (st->target.items) munmap(st->target.items, st->items_bytes);
      st->target.words = nullptr;
      st->target.items = nullptr;
      if (std::getenv("PENGK_TIMING"))
        std::cerr << "[timing] (host copies of the packed input released: " << (held >> 20) << " MiB resident)" << std::endl;
    }
  }
  if (!st->in.d_words) return nullptr;  // no records at all on this rank: the staged path handles the empty shard
  return &st->in;
}

const PackedInput* packed_input(SequenceSet* set, int W) {
  Stream* st = g_stream;
  return st && st->set == set && st->in.W == W && st->in.d_words ? &st->in : nullptr;
}

void shutdown() {
  join_starter();  // (the warm-up thread reads the context: found by tests/tools/tsan_host.sh)
  Lap lap("  ");
  if (g_stream && g_ctx) {
    if (g_stream->in.d_words) pengk_free(g_ctx, g_stream->in.d_words);
    if (g_stream->in.d_items) pengk_free(g_ctx, g_stream->in.d_items);
  }
  delete g_stream;
  g_stream = nullptr;
  lap("device copy of the packed input released");
  if (g_ctx) pengk_destroy(g_ctx);
  g_ctx = nullptr;
  lap("context destroyed (scratch buffers, streams)");
}

}  // namespace pengk_host
