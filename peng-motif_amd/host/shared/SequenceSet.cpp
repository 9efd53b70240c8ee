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
    const size_t n = getN();
    sequences_.reserve(n);
    for (size_t i = 0; i < n; ++i) {
      const int L = (int)(offs_[i + 1] - offs_[i]);
      if (single_stranded_) {
        sequences_.push_back(Sequence::view(codes_ + offs_[i], L, headers_[i]));
      } else {
        sequences_.push_back(new Sequence(codes_ + offs_[i], L, headers_[i], std::vector<int>(), false));
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

}  // namespace

void SequenceSet::readFASTA() {
  const int fd = open(path_.c_str(), O_RDONLY);
  struct stat sb;
  if (fd < 0 || fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) {
    std::cerr << "Error: Cannot open FASTA file: " << path_ << std::endl;
    exit(1);
  }
  size_t size = (size_t)sb.st_size;
  const char* text = nullptr;
  if (size) {
    text = (const char*)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (text == MAP_FAILED) {
      std::cerr << "Error: Cannot open FASTA file: " << path_ << std::endl;
      exit(1);
    }
    madvise((void*)text, size, MADV_SEQUENTIAL);
  }
  close(fd);
  const size_t mapped = size;
  // the reference's getline(...).good() loop never yields a final line without '\n'
  while (size && text[size - 1] != '\n') --size;

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
  {
    const size_t first = hdr.empty() ? size : hdr[0];
    for (size_t i = 0; i < first; ++i)
      if (text[i] != '\n') {
        std::cerr << "Error: Wrong FASTA format: " << path_ << std::endl;
        exit(1);
      }
  }
  const size_t R = hdr.size();
  hdr.push_back(size);

  // ---- 2. measure every record (sequence length, spaces) ---------------------------------------------------
  std::vector<uint32_t> len(R, 0);
  std::vector<int> bad(nt, 0);
  auto record_cut = [&](unsigned t) { return (size_t)((uint64_t)R * t / nt); };
  parallel_for(nt, [&](unsigned t) {
    for (size_t r = record_cut(t); r < record_cut(t + 1); ++r) {
      const char* p = text + hdr[r];
      const char* end = text + hdr[r + 1];
      p = (const char*)memchr(p, '\n', (size_t)(end - p)) + 1;  // skip the header line (terminated: size ends on '\n')
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
    if (b) {
      std::cerr << "Error: FASTA sequence contains space character: " << path_ << std::endl;
      exit(1);
    }

  // ---- 3. offsets of the kept records, headers -------------------------------------------------------------
  std::vector<size_t> kept;
  kept.reserve(R);
  int64_t total = 0;
  for (size_t r = 0; r < R; ++r) {
    if (len[r] == 0) {
      std::cerr << "Warning: Ignore FASTA entry without sequence: " << path_ << std::endl;
      diagnostics_ += "Warning: Ignore FASTA entry without sequence: " + path_ + "\n";
      continue;
    }
    kept.push_back(r);
    total += len[r];
    offs_.push_back(total);
    if (len[r] > maxL_) maxL_ = len[r];
    if (len[r] < minL_) minL_ = len[r];
  }
  headers_.resize(kept.size());
  codes_ = (uint8_t*)std::malloc(total ? (size_t)total : 1);
  if (!codes_) {
    std::cerr << "Error: out of memory reading " << path_ << std::endl;
    exit(1);
  }

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
  const size_t K = kept.size();
  parallel_for(nt, [&](unsigned t) {
    unsigned long* bc = counts[t].data();
    for (size_t k = (size_t)((uint64_t)K * t / nt); k < (size_t)((uint64_t)K * (t + 1) / nt); ++k) {
      const size_t r = kept[k];
      const char* p = text + hdr[r];
      const char* end = text + hdr[r + 1];
      const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
      // ">" alone: the header is the 1-based index among the kept records (reference: N+1)
      headers_[k] = (nl - p == 1) ? std::to_string(k + 1) : std::string(p + 1, (size_t)(nl - p - 1));
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
  // (:328-336).  Same stderr here.
  if (R && len[R - 1] != 0) {
    const char* p = text + hdr[R - 1];
    const char* end = text + hdr[R];
    const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
    const std::string& header = headers_.back();
    for (p = nl + 1; p < end; ++p)
      if (*p != '\n' && lut[(uint8_t)*p] == 0) {
        std::cerr << "Warning: The FASTA file contains an undefined base: " << *p << " at sequence " << header << std::endl;
        diagnostics_ += std::string("Warning: The FASTA file contains an undefined base: ") + *p + " at sequence " + header + "\n";
      }
  }
  unsigned long base_counts[4] = {0, 0, 0, 0};
  for (auto& c : counts)
    for (int i = 0; i < 4; ++i) base_counts[i] += c[i + 1];
  const unsigned long sum = base_counts[0] + base_counts[1] + base_counts[2] + base_counts[3];
  for (int i = 0; i < 4; ++i) base_freq_[i] = (float)base_counts[i] / (float)sum;
  if (text) munmap((void*)text, mapped);
}
