#include "SequenceSet.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <limits>
#include <thread>

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
  offs_.assign(1, 0);
  readFASTA();
  if (!intensityFilepath.empty()) {
    std::cerr << "Error: SequenceSet::readIntensities() is not implemented so far." << std::endl;
    exit(1);
  }
}

SequenceSet::~SequenceSet() {
  for (Sequence* s : sequences_) delete s;
  std::free(codes_);
}

std::vector<Sequence*> SequenceSet::getSequences() {
  if (!materialised_) {
    const size_t n = getLocalN();
    sequences_.reserve(n);
    for (size_t i = 0; i < n; ++i) {
      const int L = (int)(offs_[i + 1] - offs_[i]);
      if (single_stranded_) {
        sequences_.push_back(Sequence::view(codes_ + offs_[i], L, header(i)));
      } else {
        sequences_.push_back(new Sequence(codes_ + offs_[i], L, header(i), std::vector<int>(), false));
      }
    }
    materialised_ = true;
  }
  return sequences_;
}

namespace {

struct Span {
  size_t begin, end;  // record text [begin, end): header line first
};

unsigned host_threads(size_t bytes) {
  unsigned hw = std::thread::hardware_concurrency();
  if (hw == 0) hw = 1;
  if (hw > 32) hw = 32;
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

// first record start ('>' at the start of a line) at or after `from`, or `limit` when there is none
size_t next_record_start(int fd, size_t from, size_t limit, bool* io_error) {
  if (from == 0) return 0;
  if (from >= limit) return limit;
  std::vector<char> buf((size_t)1 << 20);
  size_t pos = from - 1;  // buf[0] is the byte in front of the first candidate
  while (pos + 1 < limit) {
    const size_t want = std::min(buf.size(), limit - pos);
    size_t got = 0;
    while (got < want) {
      const ssize_t k = pread(fd, buf.data() + got, want - got, (off_t)(pos + got));
      if (k <= 0) {
        *io_error = true;
        return limit;
      }
      got += (size_t)k;
    }
    const char* p = buf.data() + 1;
    const char* e = buf.data() + got;
    while (p < e && (p = (const char*)memchr(p, '>', (size_t)(e - p)))) {
      if (p[-1] == '\n') return pos + (size_t)(p - buf.data());
      ++p;
    }
    pos += got - 1;
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

SequenceShardComm g_shard_comm;

}  // namespace

void SequenceSet::setShardComm(const SequenceShardComm& comm) { g_shard_comm = comm; }
const SequenceShardComm& SequenceSet::shardComm() { return g_shard_comm; }

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

void SequenceSet::readFASTA() {
  const SequenceShardComm& sc = g_shard_comm;
  ShardSummary mine{};
  std::string last_record_warnings;
  raw_vector<uint32_t> len;  // sequence length of every record of the shard (0: dropped)
  size_t R = 0;              // records of the shard
  bool owns_file_end = false;

  // ---- this rank's part of the file: everything a single process does with the whole file ---------------------------
  auto local_pass = [&]() -> int {
  const int fd = open(path_.c_str(), O_RDONLY);
  struct stat sb;
  if (fd < 0 || fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) {
    if (fd >= 0) close(fd);
    return FASTA_OPEN;
  }
  size_t file_size = (size_t)sb.st_size;
  // the reference's getline(...).good() loop never yields a final line without '\n'
  {
    char tail[4096];
    while (file_size) {
      const size_t n = std::min(sizeof tail, file_size);
      if (pread(fd, tail, n, (off_t)(file_size - n)) != (ssize_t)n) {
        close(fd);
        return FASTA_OPEN;
      }
      size_t i = n;
      while (i && tail[i - 1] != '\n') --i;
      file_size -= n - i;
      if (i) break;
    }
  }
  // byte range of this rank, cut at record starts
  size_t lo = 0, hi = file_size;
  if (sc.world > 1) {
    bool io_error = false;
    lo = next_record_start(fd, (size_t)((unsigned __int128)file_size * sc.rank / sc.world), file_size, &io_error);
    if (sc.rank + 1 < sc.world)
      hi = next_record_start(fd, (size_t)((unsigned __int128)file_size * (sc.rank + 1) / sc.world), file_size, &io_error);
    if (io_error) {
      close(fd);
      return FASTA_OPEN;
    }
  }
  owns_file_end = hi == file_size && hi > lo;
  const size_t size = hi - lo;
  // The range is copied into an anonymous buffer on huge pages by all threads (pread) instead of mapped: mapping touches
  // every 4 KiB page of the page cache once per process, and that first touch was most of the reader's time for a
  // 2 GB input (the three passes below then also run on 2 MiB pages).
  char* text = nullptr;
  struct TextGuard {
    char*& p;
    ~TextGuard() { std::free(p); }
  } text_guard{text};
  if (size) {
    text = (char*)sequence_set_huge_alloc(size);
    if (!text) {
      close(fd);
      return FASTA_NOMEM;
    }
    const unsigned nr = host_threads(size);
    std::vector<int> failed(nr, 0);
    parallel_for(nr, [&](unsigned t) {
      size_t at = size * t / nr;
      const size_t end = size * (t + 1) / nr;
      while (at < end) {
        const ssize_t got = pread(fd, text + at, end - at, (off_t)(lo + at));
        if (got <= 0) {
          failed[t] = 1;
          return;
        }
        at += (size_t)got;
      }
    });
    for (int f : failed)
      if (f) {
        close(fd);
        return FASTA_OPEN;
      }
  }
  close(fd);

  const unsigned nt = host_threads(size);

  // ---- 1. headers: '>' at the start of a line ---------------------------------------------------------
  std::vector<std::vector<size_t>> found(nt);
  parallel_for(nt, [&](unsigned t) {
    const size_t lo = size * t / nt, hi = size * (t + 1) / nt;
    const char* p = text + lo;
    while (p < text + hi) {
      p = (const char*)memchr(p, '>', (size_t)(text + hi - p));
      if (!p) break;
      const size_t at = (size_t)(p - text);
      if (at == 0 || text[at - 1] == '\n') found[t].push_back(at);
      ++p;
    }
  });
  std::vector<size_t> hdr;
  for (auto& v : found) hdr.insert(hdr.end(), v.begin(), v.end());
  // anything but blank lines in front of the first header is a format error (reference: exit(1))
  // (only the rank that holds the head of the file can see such text: every other range starts on a header)
  {
    const size_t first = hdr.empty() ? size : hdr[0];
    for (size_t i = 0; i < first; ++i)
      if (text[i] != '\n') return FASTA_FORMAT;
  }
  R = hdr.size();
  hdr.push_back(size);

  // ---- 2. measure every record (sequence length, spaces) ---------------------------------------------------
  raw_vector<uint32_t> hlen(R);
  len.resize(R);  // both written for every record below
  std::vector<int> bad(nt, 0);
  auto record_cut = [&](unsigned t) { return (size_t)((uint64_t)R * t / nt); };
  parallel_for(nt, [&](unsigned t) {
    for (size_t r = record_cut(t); r < record_cut(t + 1); ++r) {
      const char* p = text + hdr[r];
      const char* end = text + hdr[r + 1];
      const char* h_end = (const char*)memchr(p, '\n', (size_t)(end - p));  // the header line (terminated: size ends on '\n')
      hlen[r] = (uint32_t)(h_end - p - 1);
      p = h_end + 1;
      size_t n = 0;
      while (p < end) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        const size_t l = (size_t)(nl - p);
        if (l && memchr(p, ' ', l)) bad[t] = 1;
        n += l;
        p = nl + 1;
      }
      len[r] = (uint32_t)n;
    }
  });
  for (int b : bad)
    if (b) return FASTA_SPACE;

  // ---- 3. offsets of the kept records, headers -------------------------------------------------------------
  // Two parallel passes over the records (totals per thread range, then every range writes its own entries): with
  // push_backs, one std::string per header and value-initialised arrays this step ran on one thread and was the
  // longest of the reader at 10M records.
  struct Part {
    size_t kept = 0, empty = 0;
    int64_t bases = 0;
    uint64_t hbytes = 0;
    uint32_t maxL = 0, minL = std::numeric_limits<uint32_t>::max();
  };
  std::vector<Part> part(nt);
  parallel_for(nt, [&](unsigned t) {
    Part p;
    for (size_t r = record_cut(t); r < record_cut(t + 1); ++r) {
      const uint32_t l = len[r];
      if (l == 0) {
        ++p.empty;
        continue;
      }
      ++p.kept;
      p.bases += l;
      p.hbytes += hlen[r];
      p.maxL = l > p.maxL ? l : p.maxL;
      p.minL = l < p.minL ? l : p.minL;
    }
    part[t] = p;
  });
  size_t K = 0, n_empty = 0;
  int64_t total = 0;
  uint64_t htotal = 0;
  std::vector<Part> base(nt);  // what lies in front of each thread's range
  for (unsigned t = 0; t < nt; ++t) {
    base[t].kept = K;
    base[t].bases = total;
    base[t].hbytes = htotal;
    K += part[t].kept;
    n_empty += part[t].empty;
    total += part[t].bases;
    htotal += part[t].hbytes;
    if (part[t].kept) {
      if (part[t].maxL > maxL_) maxL_ = part[t].maxL;
      if (part[t].minL < minL_) minL_ = part[t].minL;
    }
  }
  mine.records = R;
  mine.kept = K;
  mine.empty = n_empty;
  mine.bases = (uint64_t)total;
  mine.minL = minL_;
  mine.maxL = maxL_;
  raw_vector<size_t> kept(K);
  offs_.resize(K + 1);
  hdr_off_.resize(K + 1);
  offs_[0] = 0;
  hdr_off_[0] = 0;
  parallel_for(nt, [&](unsigned t) {
    size_t k = base[t].kept;
    int64_t at = base[t].bases;
    uint64_t hat = base[t].hbytes;
    for (size_t r = record_cut(t); r < record_cut(t + 1); ++r) {
      if (len[r] == 0) continue;
      kept[k] = r;
      at += len[r];
      hat += hlen[r];
      ++k;
      offs_[k] = at;
      hdr_off_[k] = hat;
    }
  });
  hdr_pool_.resize((size_t)htotal);
  codes_ = (uint8_t*)sequence_set_huge_alloc((size_t)total);
  if (!codes_) return FASTA_NOMEM;

  // ---- 4. translate ------------------------------------------------------------------------------------------------
  // A C G T (either case) -> 1 2 3 4, every other byte -> 0 (Alphabet.cpp:33-41), written as branch-free byte
  // arithmetic so that the compiler vectorises the line loop; base counts per line the same way.
  uint8_t lut[256];
  bool standard = true;  // does the arithmetic below reproduce the alphabet's table for every byte?
  for (int c = 0; c < 256; ++c) {
    lut[c] = Alphabet::getCode((char)c);
    const int u = c & 0xDF;
    standard &= lut[c] == (u == 'A') + 2 * (u == 'C') + 3 * (u == 'G') + 4 * (u == 'T');
  }
  std::vector<std::vector<unsigned long>> counts(nt, std::vector<unsigned long>(5, 0));
  parallel_for(nt, [&](unsigned t) {
    unsigned long* bc = counts[t].data();
    for (size_t k = (size_t)((uint64_t)K * t / nt); k < (size_t)((uint64_t)K * (t + 1) / nt); ++k) {
      const size_t r = kept[k];
      const char* p = text + hdr[r];
      const char* end = text + hdr[r + 1];
      const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
      // (">" alone: an empty header, read back as the 1-based index among the kept records -- reference: N+1)
      if (nl - p > 1) memcpy(hdr_pool_.data() + hdr_off_[k], p + 1, (size_t)(nl - p - 1));
      p = nl + 1;
      uint8_t* out = codes_ + offs_[k];
      while (p < end) {
        nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        const size_t l = (size_t)(nl - p);
        const uint8_t* in = (const uint8_t*)p;
        if (standard) {
          unsigned a = 0, c2 = 0, g = 0, tt = 0;
          for (size_t i = 0; i < l; ++i) {
            const uint8_t u = (uint8_t)(in[i] & 0xDF);  // upper case
            const uint8_t ia = u == 'A', ic = u == 'C', ig = u == 'G', it = u == 'T';
            out[i] = (uint8_t)(ia + 2 * ic + 3 * ig + 4 * it);
            a += ia;
            c2 += ic;
            g += ig;
            tt += it;
          }
          bc[1] += a;
          bc[2] += c2;
          bc[3] += g;
          bc[4] += tt;
        } else {
          for (size_t i = 0; i < l; ++i) {
            const uint8_t c = lut[in[i]];
            out[i] = c;
            ++bc[c];
          }
        }
        out += l;
        p = nl + 1;
      }
    }
  });
  // The reference reports undefined bases only for the LAST record of the file -- the one its reader handles behind
  // the line loop (src/shared/SequenceSet.cpp:395-405); records closed by a following header are translated silently
  // (:328-336).  Same stderr here: the rank that holds the end of the file collects the offending characters.
  if (owns_file_end && R && len[R - 1] != 0) {
    const char* p = text + hdr[R - 1];
    const char* end = text + hdr[R];
    const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
    std::string bad_chars;
    for (const char* q = nl + 1; q < end; ++q)
      if (*q != '\n' && lut[(uint8_t)*q] == 0) bad_chars += *q;
    // carried to rank 0 as: header text (empty for '>' alone), a line feed, the offending characters
    if (!bad_chars.empty()) last_record_warnings = std::string(p + 1, (size_t)(nl - p - 1)) + "\n" + bad_chars;
  }
  for (auto& c : counts)
    for (int i = 0; i < 4; ++i) mine.counts[i] += c[i + 1];
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
