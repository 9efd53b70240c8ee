// SequenceSet -- FASTA reader of the host mirror (public interface of src/shared/SequenceSet.h).
// Parsing quirks of the reference are kept (src/shared/SequenceSet.cpp:285-447): an unterminated last
// line is not seen, blank lines are skipped, a header without sequence is dropped with a warning,
// a space inside a sequence line or a sequence before any header ends the program with exit(1).
//
// Ingest is built for 10^7-record inputs (83 % of the reference's wall time at BASELINE config 3,
// SURVEY.md 8f.1): the file is read in chunks by all host cores (below); Sequence objects are only
// materialised if getSequences() is called.
#ifndef PENGK_HOST_SEQUENCESET_H_
#define PENGK_HOST_SEQUENCESET_H_

#include <stdint.h>

#include <cstdlib>
#include <memory>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "Alphabet.h"
#include "Sequence.h"

// std::vector whose resize() leaves new elements uninitialised: the reader fills its large arrays from all threads, and
// a value-initialising resize would touch (and zero) every page from one thread first
template <class T>
struct default_init_allocator : std::allocator<T> {
  template <class U>
  struct rebind {
    typedef default_init_allocator<U> other;
  };
  default_init_allocator() = default;
  template <class U>
  default_init_allocator(const default_init_allocator<U>&) {}
  // large arrays on transparent huge pages: a first touch then costs one fault per 2 MiB instead of one per 4 KiB
  // (page faults are slow inside a VM -- ~10 us each measured -- and these arrays are touched exactly once)
  T* allocate(std::size_t n) { return static_cast<T*>(huge_alloc(n * sizeof(T))); }
  void deallocate(T* p, std::size_t) { std::free(p); }
  static void* huge_alloc(std::size_t bytes);
  template <class U>
  void construct(U* p) {
    ::new (static_cast<void*>(p)) U;
  }
  template <class U, class... A>
  void construct(U* p, A&&... a) {
    ::new (static_cast<void*>(p)) U(std::forward<A>(a)...);
  }
};
void* sequence_set_huge_alloc(std::size_t bytes);  // SequenceSet.cpp: aligned_alloc + MADV_HUGEPAGE from 4 MiB on
template <class T>
void* default_init_allocator<T>::huge_alloc(std::size_t bytes) {
  void* p = sequence_set_huge_alloc(bytes);
  if (!p) throw std::bad_alloc();
  return p;
}
template <class T>
using raw_vector = std::vector<T, default_init_allocator<T>>;

// Sharded ingest (multi-GPU runs, one process per GPU): rank r of `world` reads only its byte range of the file, cut at
// record starts ('>' at the start of a line), and holds only those records.  What the reference derives from the whole
// file -- N, min / max length, base frequencies, its warnings -- is combined over the ranks through `allgather`
// (recv[r * bytes ...) = rank r's send[0 .. bytes); returns false when a rank is lost).  The reader itself stays free of
// the device library: the CLI plugs the host channel of include/pengk.h in here (Global.cpp), tests plug in files.
struct SequenceShardComm {
  int rank = 0, world = 1;
  bool (*allgather)(const void* send, void* recv, size_t bytes) = nullptr;
};

// The reader works in CHUNKS: byte ranges of a few MiB, cut at record starts, each read (pread into a buffer that stays
// in cache), parsed and translated by one host thread from start to end -- so the first chunks are complete while the
// rest of the file is still being read, and a consumer can take them from there: the CLI packs every chunk to 2 bits per
// base and sends it to the GPU as it arrives (host/device.cpp), which hides packing and upload behind the read.
struct SequenceChunk {
  uint8_t* codes = nullptr;        // byte codes of the chunk's kept records, back to back
  uint8_t* released = nullptr;     // (their address range once a chunk consumer has taken them: pages gone, freed with the set)
  raw_vector<int64_t> offs;        // n + 1 offsets into codes (offs[0] = 0)
  raw_vector<char> hdr_pool;       // headers, compact: header k = hdr_pool[hdr_off[k] .. hdr_off[k + 1])
  raw_vector<uint64_t> hdr_off;
  size_t n = 0;                    // kept records
  size_t first = 0;                // index of the chunk's first record among this rank's records (set after the read)
};
struct SequenceChunkSink {
  void* user = nullptr;
  // before the first chunk: `range_bytes` of FASTA text will arrive in `n_chunks` chunks (an upper bound on the bases)
  void (*begin)(void* user, size_t range_bytes, size_t n_chunks) = nullptr;
  // from the reader's worker threads, concurrently, in any order: chunk `index` is complete.  Returns true if the
  // consumer has taken everything it needs from the chunk's byte codes: the reader thread then gives them back at once
  // (records, headers and lengths stay; codes() / getSequences() of such a set end the program with a message).  2 GB of
  // codes given back by sixteen threads while the file is still being read cost nothing; left to the end of the process
  // they are 0.2 s of single-threaded page freeing that the caller of the program waits for.
  bool (*chunk)(void* user, size_t index, const SequenceChunk& c) = nullptr;
};

class SequenceSet {
 public:
  SequenceSet(std::string sequenceFilepath, bool single_stranded = false, std::string intensityFilepath = "");
  ~SequenceSet();

  // process-wide; set before the first SequenceSet is constructed (default: one rank, the whole file)
  static void setShardComm(const SequenceShardComm& comm);
  static const SequenceShardComm& shardComm();
  // sum over the ranks of n 64-bit counters (in place) through shardComm(); exits(1) when a rank is lost
  static void allreduceSum(long long* values, size_t n);
  // consumer of the chunks of the NEXT set that is constructed (cleared by that constructor)
  static void setChunkSink(const SequenceChunkSink& sink);

  std::string getSequenceFilepath() { return path_; }
  // the warnings this set's constructor wrote to stderr (a caller that re-uses the set where the reference reads the
  // file again replays them)
  const std::string& diagnostics() const { return diagnostics_; }
  std::vector<Sequence*> getSequences();  // materialises views on first use
  size_t getN() { return n_global_; }                 // records of the whole file (all ranks)
  size_t getLocalN() { return n_local_; }              // records this rank holds
  size_t getLocalBase() { return k_base_; }            // number of records in front of this rank's first one
  unsigned int getMinL() { return minL_; }
  unsigned int getMaxL() { return maxL_; }
  float* getBaseFrequencies() { return base_freq_; }

  // this rank's records, chunk by chunk, in file order
  bool codesReleased() const { return codes_released_; }  // a chunk consumer took the byte codes (see SequenceChunkSink)
  size_t nChunks() const { return chunks_.size(); }
  const SequenceChunk& chunk(size_t i) const { return chunks_[i]; }
  // contiguous copy, made on first use (tools and callers that want one array): codes of local record i are
  // codes()[offsets()[i] .. offsets()[i+1])
  const uint8_t* codes() { return flatten(), flat_codes_; }
  const int64_t* offsets() { return flatten(), flat_offs_.data(); }

 private:
  void readFASTA(const SequenceChunkSink& sink);
  void flatten();
  std::string header(size_t k) const;  // of local record k
  std::string path_;
  std::string diagnostics_;
  bool single_stranded_;
  std::vector<SequenceChunk> chunks_;
  bool codes_released_ = false;
  uint8_t* flat_codes_ = nullptr;
  raw_vector<int64_t> flat_offs_;
  size_t n_global_ = 0, n_local_ = 0, k_base_ = 0;
  std::vector<Sequence*> sequences_;
  bool materialised_ = false;
  unsigned int minL_, maxL_;
  float base_freq_[4];
};

#endif
