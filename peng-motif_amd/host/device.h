// device.h -- the one pengk context of the process and RAII device buffers for the host mirror.
// Every hot loop of BasePattern / IUPACPattern / Peng goes through include/pengk.h; there is no CPU
// fallback: a failing call prints pengk_last_error() and exits like the reference's error paths do.
#ifndef PENGK_HOST_DEVICE_H_
#define PENGK_HOST_DEVICE_H_

#include <chrono>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <iostream>

#include "pengk.h"

class SequenceSet;

namespace pengk_host {

pengk_ctx* context();           // created on first use on Global::device
void start_context();           // optional: begin creating it on a helper thread; context() waits for it
void shutdown();
void check(int rc, const char* what);

// Multi-GPU runs: one process per GPU, started by a launcher that sets RANK / WORLD_SIZE / LOCAL_RANK (torchrun,
// mpirun wrappers).  Read from the environment BEFORE any GPU call; the context then lives on device LOCAL_RANK and
// carries an RCCL communicator (pengk_comm_init_env).  Without a launcher: rank 0 of 1.
int rank();
int world();
bool launched();  // WORLD_SIZE is set (even to 1): the run goes through the communicator
// world > 1: opens the host channel of the job (include/pengk.h) and makes the FASTA reader shard by it -- rank r
// reads, translates and holds only its byte range of every input file.  Call before the first SequenceSet.
void start_sharded_ingest();
// launched runs: waits for every rank, then releases the communicator and the host channel (call before leaving)
void finish_ranks();

// The input set of the run, packed while it is read: the FASTA reader hands every finished chunk of records to the
// packer on its own worker threads (SequenceChunkSink, shared/SequenceSet.h), which packs it straight into one pair of
// host buffers that collect all chunks (pengk_pack_append); an uploader thread sends the stream to the device in 32 MiB
// pieces as they fill up -- the device context is still starting when the first ones are ready -- so that packing and
// upload hide behind the read (src/main.cpp:18-84 does these steps one after the other).
// The packer's background counters come with it: when the input set doubles as the background set (the default,
// src/Global.cpp:66-75) the model is built from them and no separate pass over the sequences is needed.
struct PackedInput {
  int W = 0, item_windows = 0;
  uint64_t* d_words = nullptr;  // device buffers that collected the chunks (owned here, released with the context)
  uint64_t* d_items = nullptr;
  uint64_t n_words = 0, n_items = 0, n_windows = 0, max_bin_bound = 0;
  int all_whole = 1;
  long long bg_counts[84] = {0};  // this rank's records (BaMM ids), as BackgroundModel counts them
};
void begin_streaming_pack(int W);                              // the NEXT SequenceSet constructed is packed as it is read
const PackedInput* finish_streaming_pack(::SequenceSet* set);    // waits for the last upload; nullptr if nothing was packed
const PackedInput* packed_input(::SequenceSet* set, int W);      // of a set that went through the two calls above, else nullptr

// PENGK_TIMING=1: wall-clock report of sub-phases on stderr (stdout stays the reference's trace)
struct Lap {
  using clk = std::chrono::steady_clock;
  const char* indent;
  bool on = std::getenv("PENGK_TIMING") != nullptr;
  clk::time_point last = clk::now();
  explicit Lap(const char* ind = "  ") : indent(ind) {}
  void operator()(const char* what) {
    if (!on) return;
    const auto now = clk::now();
    std::cerr << "[timing] " << indent << what << ": " << std::chrono::duration<double>(now - last).count() << " s" << std::endl;
    last = now;
  }
};

template <class T>
class DeviceBuffer {
 public:
  DeviceBuffer() = default;
  explicit DeviceBuffer(size_t n) { resize(n); }
  ~DeviceBuffer() { release(); }
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  void resize(size_t n) {
    release();
    n_ = n;
    void* p = nullptr;
    check(pengk_malloc(context(), (n ? n : 1) * sizeof(T), &p), "pengk_malloc");
    p_ = (T*)p;
  }
  void release() {
    if (p_) check(pengk_free(context(), p_), "pengk_free");
    p_ = nullptr;
    n_ = 0;
  }
  void upload(const T* h, size_t n) { check(pengk_memcpy_h2d(context(), p_, h, n * sizeof(T)), "pengk_memcpy_h2d"); }
  void download(T* h, size_t n, size_t offset = 0) const {
    check(pengk_memcpy_d2h(context(), h, p_ + offset, n * sizeof(T)), "pengk_memcpy_d2h");
  }
  T* get() const { return p_; }
  size_t size() const { return n_; }

 private:
  T* p_ = nullptr;
  size_t n_ = 0;
};

}  // namespace pengk_host
#endif
