#include "SequenceSet.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <limits>
#include <thread>
#include <vector>

void* sequence_set_huge_alloc(std::size_t bytes) {
  if (bytes == 0) bytes = 1;
  const std::size_t huge = (std::size_t)2 << 20;
  if (bytes < 2 * huge) return std::malloc(bytes);
  const std::size_t rounded = (bytes + huge - 1) / huge * huge;
  void* p = nullptr;
  if (posix_memalign(&p, huge, rounded) != 0) p = nullptr;
  // advisory: without THP these are ordinary pages.  PENGK_NO_HUGEPAGES=1 skips it (on a host whose memory is so
  // fragmented that the kernel has to compact before it can hand out 2 MiB pages, the first touch stalls instead)
  static const bool want = std::getenv("PENGK_NO_HUGEPAGES") == nullptr;
  if (p && want) madvise(p, rounded, MADV_HUGEPAGE);
  return p;
}

namespace {

unsigned host_threads(size_t bytes) {
  unsigned hw = std::thread::hardware_concurrency();
  if (hw == 0) hw = 1;
  if (hw > 32) hw = 32;
  if (const char* e = std::getenv("PENGK_READ_THREADS")) {  // tests: force a thread count
    const int v = std::atoi(e);
    if (v >= 1 && v <= 64) return (unsigned)v;
  }
  return bytes < (1u << 22) ? 1u : hw;
}

template <class F>
void parallel_for(unsigned nt, F&& f) {
  struct Joiner {  // started threads are joined on every exit path (a throwing thread start must not terminate())
    std::vector<std::thread> th;
    ~Joiner() {
      for (auto& x : th)
        if (x.joinable()) x.join();
    }
  } j;
  for (unsigned t = 1; t < nt; ++t) j.th.emplace_back(f, t);
  f(0u);
}

bool pread_all(int fd, char* dst, size_t n, size_t at) {
  while (n) {
    const ssize_t k = pread(fd, dst, n, (off_t)at);
    if (k <= 0) return false;
    dst += k;
    at += (size_t)k;
    n -= (size_t)k;
  }
  return true;
}

// first record start ('>' at the start of a line) at or after `from`, or `limit` when there is none
size_t next_record_start(int fd, size_t from, size_t limit, bool* io_error) {
  if (from == 0) return 0;
  if (from >= limit) return limit;
  std::vector<char> buf;
  size_t pos = from - 1;  // buf[0] is the byte in front of the first candidate
  size_t step = (size_t)16 << 10;
  while (pos + 1 < limit) {
    buf.resize(step);
    const size_t got = std::min(buf.size(), limit - pos);
    if (!pread_all(fd, buf.data(), got, pos)) {
      *io_error = true;
      return limit;
    }
    const char* p = buf.data() + 1;
    const char* e = buf.data() + got;
    while (p < e && (p = (const char*)memchr(p, '>', (size_t)(e - p)))) {
      if (p[-1] == '\n') return pos + (size_t)(p - buf.data());
      ++p;
    }
    pos += got - 1;
    step = (size_t)1 << 20;
  }
  return limit;
}

// what a rank reports about its shard (fixed size: one allgather combines the ranks)
struct ShardSummary {
  uint64_t error;    // 0 or a FASTA_* code below
  uint64_t records;  // headers in the shard, with or without sequence
  uint64_t kept, empty, bases, minL, maxL;
  uint64_t counts[4];
  uint64_t warn_bytes;  // the last record's undefined-base warnings (owner only)
};
enum { FASTA_OK = 0, FASTA_OPEN = 1, FASTA_FORMAT = 2, FASTA_SPACE = 3, FASTA_NOMEM = 4 };

// what a worker reports about one chunk
struct ChunkStat {
  int error = FASTA_OK;
  size_t records = 0, empty = 0;
  int64_t bases = 0;
  uint32_t maxL = 0, minL = std::numeric_limits<uint32_t>::max();
  unsigned long counts[4] = {0, 0, 0, 0};
  std::string last_warnings;  // last chunk of the file only
};

SequenceShardComm g_shard_comm;
SequenceChunkSink g_sink;

constexpr size_t CHUNK_BYTES = (size_t)8 << 20;

// One chunk, start to end, on the calling thread: text [s, e) of the file (starts on a header unless s == 0, ends in
// front of a header or at the end of the range) -> kept records as byte codes.
// Parsing quirks of the reference (src/shared/SequenceSet.cpp:285-447): blank lines are skipped, a header without
// sequence is dropped, a space inside a sequence line or text in front of the first header is an error.
void read_chunk(int fd, size_t s, size_t e, bool file_end, const uint8_t* lut, bool standard, std::vector<char>& text,
                std::vector<size_t>& hdr, std::vector<uint32_t>& len, SequenceChunk& out, ChunkStat& st) {
  const size_t size = e - s;
  if (text.size() < size) text.resize(size + size / 8);
  if (size && !pread_all(fd, text.data(), size, s)) {
    st.error = FASTA_OPEN;
    return;
  }
  const char* t = text.data();
  // ---- headers: '>' at the start of a line ----------------------------------------------------------------
  hdr.clear();
  for (const char* p = t; p < t + size;) {
    p = (const char*)memchr(p, '>', (size_t)(t + size - p));
    if (!p) break;
    const size_t at = (size_t)(p - t);
    if (at == 0 || t[at - 1] == '\n') hdr.push_back(at);
    ++p;
  }
  // anything but blank lines in front of the first header is a format error (reference: exit(1)); only the chunk at
  // the head of the file can hold such text -- every other chunk starts on a header
  {
    const size_t first = hdr.empty() ? size : hdr[0];
    for (size_t i = 0; i < first; ++i)
      if (t[i] != '\n') {
        st.error = FASTA_FORMAT;
        return;
      }
  }
  const size_t R = hdr.size();
  hdr.push_back(size);
  st.records = R;
  // ---- measure every record (sequence length, spaces) --------------------------------------------------------
  len.resize(R);
  size_t K = 0;
  uint64_t hbytes = 0;
  for (size_t r = 0; r < R; ++r) {
    const char* p = t + hdr[r];
    const char* end = t + hdr[r + 1];
    const char* h_end = (const char*)memchr(p, '\n', (size_t)(end - p));  // the header line (terminated: ranges end on '\n')
    const size_t hl = (size_t)(h_end - p - 1);
    p = h_end + 1;
    size_t n = 0;
    while (p < end) {
      const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
      const size_t l = (size_t)(nl - p);
      if (l && memchr(p, ' ', l)) {
        st.error = FASTA_SPACE;
        return;
      }
      n += l;
      p = nl + 1;
    }
    len[r] = (uint32_t)n;
    if (n == 0) {
      ++st.empty;
      continue;
    }
    ++K;
    st.bases += (int64_t)n;
    hbytes += hl;
    if (n > st.maxL) st.maxL = (uint32_t)n;
    if (n < st.minL) st.minL = (uint32_t)n;
  }
  // ---- translate ------------------------------------------------------------------------------------------------
  // A C G T (either case) -> 1 2 3 4, every other byte -> 0 (Alphabet.cpp:33-41), written as branch-free byte
  // arithmetic so that the compiler vectorises the line loop; base counts per line the same way.
  out.n = K;
  out.offs.resize(K + 1);
  out.hdr_off.resize(K + 1);
  out.hdr_pool.resize((size_t)hbytes);
  out.codes = (uint8_t*)sequence_set_huge_alloc((size_t)st.bases);
  if (!out.codes) {
    st.error = FASTA_NOMEM;
    return;
  }
  out.offs[0] = 0;
  out.hdr_off[0] = 0;
  size_t k = 0;
  for (size_t r = 0; r < R; ++r) {
    if (len[r] == 0) continue;
    const char* p = t + hdr[r];
    const char* end = t + hdr[r + 1];
    const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
    // (">" alone: an empty header, read back as the 1-based index among the kept records -- reference: N+1)
    const size_t hl = (size_t)(nl - p - 1);
    if (hl) memcpy(out.hdr_pool.data() + out.hdr_off[k], p + 1, hl);
    out.hdr_off[k + 1] = out.hdr_off[k] + hl;
    out.offs[k + 1] = out.offs[k] + len[r];
    uint8_t* dst = out.codes + out.offs[k];
    ++k;
    p = nl + 1;
    while (p < end) {
      nl = (const char*)memchr(p, '\n', (size_t)(end - p));
      const size_t l = (size_t)(nl - p);
      const uint8_t* in = (const uint8_t*)p;
      if (standard) {
        unsigned a = 0, c2 = 0, g = 0, tt = 0;
        for (size_t i = 0; i < l; ++i) {
          const uint8_t u = (uint8_t)(in[i] & 0xDF);  // upper case
          const uint8_t ia = u == 'A', ic = u == 'C', ig = u == 'G', it = u == 'T';
          dst[i] = (uint8_t)(ia + 2 * ic + 3 * ig + 4 * it);
          a += ia;
          c2 += ic;
          g += ig;
          tt += it;
        }
        st.counts[0] += a;
        st.counts[1] += c2;
        st.counts[2] += g;
        st.counts[3] += tt;
      } else {
        for (size_t i = 0; i < l; ++i) {
          const uint8_t c = lut[in[i]];
          dst[i] = c;
          if (c) ++st.counts[c - 1];
        }
      }
      dst += l;
      p = nl + 1;
    }
  }
  // The reference reports undefined bases only for the LAST record of the file -- the one its reader handles behind
  // the line loop (src/shared/SequenceSet.cpp:395-405); records closed by a following header are translated silently
  // (:328-336).  Same stderr here: the chunk that holds the end of the file collects the offending characters.
  if (file_end && R && len[R - 1] != 0) {
    const char* p = t + hdr[R - 1];
    const char* end = t + hdr[R];
    const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
    std::string bad_chars;
    for (const char* q = nl + 1; q < end; ++q)
      if (*q != '\n' && lut[(uint8_t)*q] == 0) bad_chars += *q;
    // carried to rank 0 as: header text (empty for '>' alone), a line feed, the offending characters
    if (!bad_chars.empty()) st.last_warnings = std::string(p + 1, (size_t)(nl - p - 1)) + "\n" + bad_chars;
  }
}

}  // namespace

void SequenceSet::setShardComm(const SequenceShardComm& comm) { g_shard_comm = comm; }
const SequenceShardComm& SequenceSet::shardComm() { return g_shard_comm; }
void SequenceSet::setChunkSink(const SequenceChunkSink& sink) { g_sink = sink; }

void SequenceSet::allreduceSum(long long* values, size_t n) {
  const SequenceShardComm& sc = g_shard_comm;
  if (sc.world <= 1 || n == 0) return;
  std::vector<long long> all((size_t)sc.world * n);
  if (!sc.allgather || !sc.allgather(values, all.data(), n * sizeof(long long))) {
    std::cerr << "Error: lost a rank while combining the sequence shards" << std::endl;
    exit(1);
  }
  for (size_t i = 0; i < n; ++i) {
    long long sum = 0;
    for (int r = 0; r < sc.world; ++r) sum += all[(size_t)r * n + i];
    values[i] = sum;
  }
}

SequenceSet::SequenceSet(std::string sequenceFilepath, bool single_stranded, std::string intensityFilepath) {
  if (Alphabet::getSize() == 0) {
    std::cerr << "Error: Initialize Alphabet before constructing a SequenceSet" << std::endl;
    exit(-1);
  }
  path_ = sequenceFilepath;
  single_stranded_ = single_stranded;
  minL_ = std::numeric_limits<int>::max();
  maxL_ = 0;
  for (float& f : base_freq_) f = 0.f;
  const SequenceChunkSink sink = g_sink;  // for this set only
  g_sink = SequenceChunkSink();
  readFASTA(sink);
  if (!intensityFilepath.empty()) {
    std::cerr << "Error: SequenceSet::readIntensities() is not implemented so far." << std::endl;
    exit(1);
  }
}

SequenceSet::~SequenceSet() {
  for (Sequence* s : sequences_) delete s;
  for (SequenceChunk& c : chunks_) {
    std::free(c.codes);
    std::free(c.released);
  }
  std::free(flat_codes_);
}

std::string SequenceSet::header(size_t k) const {
  // the chunk that holds local record k
  size_t lo = 0, hi = chunks_.size();
  while (hi - lo > 1) {
    const size_t mid = (lo + hi) / 2;
    if (chunks_[mid].first <= k) lo = mid;
    else hi = mid;
  }
  const SequenceChunk& c = chunks_[lo];
  const size_t j = k - c.first;
  const uint64_t b = c.hdr_off[j], e = c.hdr_off[j + 1];
  return e == b ? std::to_string(k_base_ + k + 1) : std::string(c.hdr_pool.data() + b, (size_t)(e - b));
}

static void codes_gone(const std::string& path) {
  std::cerr << "Error: the sequences of " << path << " were handed to the device while the file was read and are no longer "
            << "held on the host (PENGK_NO_STREAMING=1 keeps them)" << std::endl;
  exit(1);
}

std::vector<Sequence*> SequenceSet::getSequences() {
  if (codes_released_) codes_gone(path_);
  if (!materialised_) {
    sequences_.reserve(n_local_);
    for (const SequenceChunk& c : chunks_)
      for (size_t j = 0; j < c.n; ++j) {
        const int L = (int)(c.offs[j + 1] - c.offs[j]);
        if (single_stranded_) {
          sequences_.push_back(Sequence::view(c.codes + c.offs[j], L, header(c.first + j)));
        } else {
          sequences_.push_back(new Sequence(c.codes + c.offs[j], L, header(c.first + j), std::vector<int>(), false));
        }
      }
    materialised_ = true;
  }
  return sequences_;
}

void SequenceSet::flatten() {
  if (flat_codes_ || flat_offs_.size()) return;
  if (codes_released_) codes_gone(path_);
  int64_t total = 0;
  for (const SequenceChunk& c : chunks_) total += c.n ? c.offs[c.n] : 0;
  flat_codes_ = (uint8_t*)sequence_set_huge_alloc((size_t)total);
  flat_offs_.resize(n_local_ + 1);
  if (!flat_codes_) {
    std::cerr << "Error: out of memory reading " << path_ << std::endl;
    exit(1);
  }
  int64_t at = 0;
  size_t k = 0;
  flat_offs_[0] = 0;
  for (const SequenceChunk& c : chunks_) {
    if (!c.n) continue;
    memcpy(flat_codes_ + at, c.codes, (size_t)c.offs[c.n]);
    for (size_t j = 0; j < c.n; ++j) flat_offs_[++k] = at + c.offs[j + 1];
    at += c.offs[c.n];
  }
}

void SequenceSet::readFASTA(const SequenceChunkSink& sink) {
  const SequenceShardComm& sc = g_shard_comm;
  ShardSummary mine{};
  std::string last_record_warnings;

  // ---- this rank's part of the file: everything a single process does with the whole file ---------------------------
  auto local_pass = [&]() -> int {
    int fd = open(path_.c_str(), O_RDONLY);
    struct FdGuard {
      int& fd;
      ~FdGuard() {
        if (fd >= 0) close(fd);
      }
    } fd_guard{fd};
    struct stat sb;
    if (fd < 0 || fstat(fd, &sb) != 0) return FASTA_OPEN;
    if (!S_ISREG(sb.st_mode)) {
      // The reference reads through std::ifstream + getline (src/shared/SequenceSet.cpp:285-300): a directory opens and
      // yields no line -- an empty set, exit code 0 --, a pipe or a device is read to its end.  Here: a directory is an
      // empty file; anything else that is not a regular file is spooled into an anonymous memory file first (one process
      // only: several ranks cannot share a pipe).
      if (S_ISDIR(sb.st_mode)) {
        sb.st_size = 0;
      } else {
        if (sc.world > 1) return FASTA_OPEN;
        const int spool = memfd_create("pengk_fasta", MFD_CLOEXEC);
        if (spool < 0) return FASTA_OPEN;
        std::vector<char> buf(1u << 20);
        bool ok = true;
        for (;;) {
          const ssize_t n = read(fd, buf.data(), buf.size());
          if (n < 0 && errno == EINTR) continue;
          if (n < 0) ok = false;
          if (n <= 0) break;
          for (ssize_t done = 0; done < n;) {
            const ssize_t w = write(spool, buf.data() + done, (size_t)(n - done));
            if (w < 0 && errno == EINTR) continue;
            if (w <= 0) {
              ok = false;
              break;
            }
            done += w;
          }
          if (!ok) break;
        }
        close(fd);
        fd = spool;
        if (!ok || fstat(fd, &sb) != 0) return FASTA_OPEN;
      }
    }
    size_t file_size = (size_t)sb.st_size;
    // the reference's getline(...).good() loop never yields a final line without '\n'
    {
      char tail[4096];
      while (file_size) {
        const size_t n = std::min(sizeof tail, file_size);
        if (!pread_all(fd, tail, n, file_size - n)) return FASTA_OPEN;
        size_t i = n;
        while (i && tail[i - 1] != '\n') --i;
        file_size -= n - i;
        if (i) break;
      }
    }
    // byte range of this rank, cut at record starts
    size_t lo = 0, hi = file_size;
    bool io_error = false;
    if (sc.world > 1) {
      lo = next_record_start(fd, (size_t)((unsigned __int128)file_size * sc.rank / sc.world), file_size, &io_error);
      if (sc.rank + 1 < sc.world)
        hi = next_record_start(fd, (size_t)((unsigned __int128)file_size * (sc.rank + 1) / sc.world), file_size, &io_error);
      if (io_error) return FASTA_OPEN;
    }
    const bool owns_file_end = hi == file_size && hi > lo;
    const size_t range = hi - lo;

    // chunks of the range; every worker finds the record starts that bound ITS chunk (two short reads), reads the chunk
    // into its own buffer -- a few MiB that stay in cache, instead of a file-sized buffer whose pages are touched once
    // by the copy and once by each pass -- and carries it through all stages
    const unsigned nt_all = host_threads(range);
    size_t C = nt_all == 1 ? 1 : std::max<size_t>(1, range / CHUNK_BYTES);
    if (const char* e = std::getenv("PENGK_READ_CHUNKS")) {  // tests: force a chunk count
      const long v = std::atol(e);
      if (v >= 1 && v <= 65536) C = (size_t)v;
    }
    if (C > range) C = range ? range : 1;
    const unsigned nt = (unsigned)std::min<size_t>(nt_all, C);
    chunks_.assign(C, SequenceChunk());
    std::vector<ChunkStat> stat(C);

    uint8_t lut[256];
    bool standard = true;  // does the arithmetic of read_chunk reproduce the alphabet's table for every byte?
    for (int c = 0; c < 256; ++c) {
      lut[c] = Alphabet::getCode((char)c);
      const int u = c & 0xDF;
      standard &= lut[c] == (u == 'A') + 2 * (u == 'C') + 3 * (u == 'G') + 4 * (u == 'T');
    }
    if (sink.begin) sink.begin(sink.user, range, C);
    std::atomic<size_t> next{0};
    std::atomic<bool> released{false};
    parallel_for(nt, [&](unsigned) {
      std::vector<char> text;
      std::vector<size_t> hdr;
      std::vector<uint32_t> len;
      for (;;) {
        const size_t c = next.fetch_add(1);
        if (c >= C) return;
        bool io = false;
        const size_t a = lo + (size_t)((unsigned __int128)range * c / C), b = lo + (size_t)((unsigned __int128)range * (c + 1) / C);
        const size_t s = c == 0 ? lo : std::max(lo, next_record_start(fd, a, hi, &io));
        const size_t e = c + 1 == C ? hi : std::max(lo, next_record_start(fd, b, hi, &io));
        if (io) {
          stat[c].error = FASTA_OPEN;
        } else {
          // (the chunk that reaches the end of the file holds its last record: a record longer than a chunk leaves the
          // chunks behind it empty)
          read_chunk(fd, s, e, owns_file_end && e == hi && s < e, lut, standard, text, hdr, len, chunks_[c], stat[c]);
        }
        // (a chunk with an error ends the run after the read: every chunk is still looked at, so that the error
        // reported is the first one in file order)
        if (!stat[c].error && sink.chunk && sink.chunk(sink.user, c, chunks_[c])) {
          // the consumer is done with the codes: their pages go back now (madvise: the address range itself stays until
          // the set dies -- unmapping takes the process's mapping lock for writing and stalls the other readers' page
          // faults: +0.05 s on a 2 GB file)
          const size_t huge = (size_t)2 << 20, bytes = (size_t)stat[c].bases;
          if (bytes >= 2 * huge) {
            madvise(chunks_[c].codes, (bytes + huge - 1) / huge * huge, MADV_DONTNEED);
            chunks_[c].released = chunks_[c].codes;
          } else {
            std::free(chunks_[c].codes);
          }
          chunks_[c].codes = nullptr;
          released.store(true, std::memory_order_relaxed);
        }
      }
    });
    codes_released_ = released.load();
    // the first error in file order is the one a single reader would have met
    for (size_t c = 0; c < C; ++c)
      if (stat[c].error) return stat[c].error;
    mine.minL = minL_;
    for (size_t c = 0; c < C; ++c) {
      chunks_[c].first = (size_t)mine.kept;
      mine.records += stat[c].records;
      mine.kept += chunks_[c].n;
      mine.empty += stat[c].empty;
      mine.bases += (uint64_t)stat[c].bases;
      if (chunks_[c].n) {
        mine.maxL = std::max<uint64_t>(mine.maxL, stat[c].maxL);
        mine.minL = std::min<uint64_t>(mine.minL, stat[c].minL);
      }
      for (int i = 0; i < 4; ++i) mine.counts[i] += stat[c].counts[i];
    }
    // the undefined bases of the file's last record: with the chunk that holds it (the last one with a record)
    for (size_t c = C; c-- > 0;)
      if (stat[c].records) {
        last_record_warnings = stat[c].last_warnings;
        break;
      }
    n_local_ = (size_t)mine.kept;
    return FASTA_OK;
  };  // local_pass

  mine.error = (uint64_t)local_pass();
  mine.warn_bytes = last_record_warnings.size();

  // ---- combine the shards --------------------------------------------------------------------------------------------
  std::vector<ShardSummary> all((size_t)sc.world);
  if (sc.world > 1) {
    if (!sc.allgather || !sc.allgather(&mine, all.data(), sizeof mine)) {
      std::cerr << "Error: lost a rank while reading " << path_ << std::endl;
      exit(1);
    }
  } else {
    all[0] = mine;
  }
  for (const ShardSummary& a : all)  // the first error in file order is the one a single reader would have met
    if (a.error) {
      if (sc.rank == 0) switch ((int)a.error) {
          case FASTA_FORMAT: std::cerr << "Error: Wrong FASTA format: " << path_ << std::endl; break;
          case FASTA_SPACE: std::cerr << "Error: FASTA sequence contains space character: " << path_ << std::endl; break;
          case FASTA_NOMEM: std::cerr << "Error: out of memory reading " << path_ << std::endl; break;
          default: std::cerr << "Error: Cannot open FASTA file: " << path_ << std::endl;
        }
      exit(1);
    }
  uint64_t n_empty_all = 0, base_counts[4] = {0, 0, 0, 0};
  int owner = -1;  // the rank that holds the last record of the file
  n_global_ = 0;
  for (int r = 0; r < sc.world; ++r) {
    const ShardSummary& a = all[(size_t)r];
    if (r == sc.rank) k_base_ = n_global_;
    n_global_ += a.kept;
    n_empty_all += a.empty;
    if (a.kept) {
      if (a.maxL > maxL_) maxL_ = (unsigned)a.maxL;
      if (a.minL < minL_) minL_ = (unsigned)a.minL;
    }
    for (int i = 0; i < 4; ++i) base_counts[i] += a.counts[i];
    if (a.records) owner = r;
  }
  // warnings: rank 0 speaks for the file, in file order (identical lines: the reference prints one per empty record as
  // it meets it; the last record's undefined bases come last)
  if (owner >= 0 && owner != sc.rank) last_record_warnings.clear();  // (an earlier rank's last record is not the file's)
  if (sc.world > 1 && owner >= 0 && all[(size_t)owner].warn_bytes) {
    const size_t n = (size_t)all[(size_t)owner].warn_bytes;
    std::vector<char> send(n, 0), recv(n * (size_t)sc.world);
    if (sc.rank == owner) memcpy(send.data(), last_record_warnings.data(), n);
    if (!sc.allgather(send.data(), recv.data(), n)) {
      std::cerr << "Error: lost a rank while reading " << path_ << std::endl;
      exit(1);
    }
    last_record_warnings.assign(recv.data() + (size_t)owner * n, n);
  }
  if (sc.rank == 0) {
    for (uint64_t i = 0; i < n_empty_all; ++i) {
      std::cerr << "Warning: Ignore FASTA entry without sequence: " << path_ << std::endl;
      diagnostics_ += "Warning: Ignore FASTA entry without sequence: " + path_ + "\n";
    }
    if (!last_record_warnings.empty()) {
      // the last record of the file is the last kept one: '>' alone reads as its 1-based index among the kept records
      const size_t cut = last_record_warnings.find('\n');
      std::string header = last_record_warnings.substr(0, cut);
      if (header.empty()) header = std::to_string(n_global_);
      for (size_t i = cut + 1; i < last_record_warnings.size(); ++i) {
        const char c = last_record_warnings[i];
        std::cerr << "Warning: The FASTA file contains an undefined base: " << c << " at sequence " << header << std::endl;
        diagnostics_ += std::string("Warning: The FASTA file contains an undefined base: ") + c + " at sequence " + header + "\n";
      }
    }
  }
  const uint64_t sum = base_counts[0] + base_counts[1] + base_counts[2] + base_counts[3];
  for (int i = 0; i < 4; ++i) base_freq_[i] = (float)base_counts[i] / (float)sum;
}
