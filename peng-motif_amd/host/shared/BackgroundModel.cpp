// Counting follows src/shared/Sequence.cpp:28-33 + src/shared/BackgroundModel.cpp:60-84: with an invalid
// base among the (up to) 9 positions [i-8, i] only an all-zero (k+1)-mer (invalid bases count as digit 0)
// is counted; otherwise every (k+1)-mer ending at i >= k is.  calculateV follows :490-530 in float32.
#include "BackgroundModel.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <thread>

void BackgroundModel::init(SequenceSet& sequenceSet, int order, std::vector<float> alpha, bool interpolate) {
  if (order < 0 || order > 2 || (int)alpha.size() < order + 1) {
    std::cerr << "Error: background model order " << order << " is not supported (0..2)" << std::endl;
    exit(-1);
  }
  const std::string path = sequenceSet.getSequenceFilepath();
  const size_t slash = path.find_last_of('/');
  name_ = slash == std::string::npos ? path : path.substr(slash + 1);
  const size_t dot = name_.find_last_of('.');
  if (dot != std::string::npos) name_ = name_.substr(0, dot);
  K_ = order;
  A_ = alpha;
  interpolate_ = interpolate;
  for (int k = 0; k < 3; ++k) {
    n_[k] = new long long[1 << (2 * (k + 1))]();
    v_[k] = new float[1 << (2 * (k + 1))]();
  }
}

BackgroundModel::BackgroundModel(SequenceSet& sequenceSet, int order, std::vector<float> alpha, bool interpolate,
                                 const long long* counts84) {
  init(sequenceSet, order, alpha, interpolate);
  for (int k = 0, at = 0; k < 3; at += 1 << (2 * (k + 1)), ++k)
    for (int y = 0; y < (1 << (2 * (k + 1))); ++y) n_[k][y] = counts84[at + y];
  for (int k = 0; k < 3; ++k) SequenceSet::allreduceSum(n_[k], (size_t)1 << (2 * (k + 1)));
  calculateV();
}

BackgroundModel::BackgroundModel(SequenceSet& sequenceSet, int order, std::vector<float> alpha, bool interpolate,
                                 std::vector<std::vector<int>>, std::vector<int>) {
  init(sequenceSet, order, alpha, interpolate);
  // count over this rank's chunks of records (the counters are summed over the ranks below), chunks dealt to host threads
  if (sequenceSet.codesReleased()) {  // (a set whose chunks went to the device comes with the packer's counters: the other constructor)
    std::cerr << "Error: background model: the sequences are no longer held on the host" << std::endl;
    exit(1);
  }
  const size_t NC = sequenceSet.nChunks();
  unsigned nt = std::thread::hardware_concurrency();
  if (nt == 0) nt = 1;
  if (nt > 32) nt = 32;
  if (NC < nt) nt = NC ? (unsigned)NC : 1u;
  std::vector<std::vector<long long>> part(nt, std::vector<long long>(168, 0));  // [0,84) fast path, [84,168) exact path
  std::atomic<size_t> next{0};
  auto work = [&](unsigned t) {
    long long* c[3] = {part[t].data(), part[t].data() + 4, part[t].data() + 20};
    for (;;) {
      const size_t ci = next.fetch_add(1);
      if (ci >= NC) return;
      const SequenceChunk& ch = sequenceSet.chunk(ci);
      const uint8_t* codes = ch.codes;
      const int64_t* offs = ch.offs.data();
    for (size_t s = 0; s < ch.n; ++s) {
      const uint8_t* seq = codes + offs[s];
      const int64_t L = offs[s + 1] - offs[s];
      if (!std::memchr(seq, 0, (size_t)L)) {
        // no invalid base: one 3-mer bin per base; 1- and 2-mer counts follow as marginals below, with the
        // first base / first 2-mer of the sequence added (they end no 3-mer)
        unsigned y = 0;
        if (L >= 1) {
          y = seq[0] - 1u;
          ++c[0][y];
        }
        if (L >= 2) {
          y = (y << 2) | (seq[1] - 1u);
          ++c[1][y];
        }
        long long* c3 = c[2];
        for (int64_t i = 2; i < L; ++i) {
          y = ((y << 2) | (seq[i] - 1u)) & 63u;
          ++c3[y];
        }
        continue;
      }
      // exact rule with invalid bases: all orders counted directly; marked so the marginals skip them
      long long* d[3] = {part[t].data() + 84, part[t].data() + 88, part[t].data() + 104};
      unsigned digits = 0, invalid = 0;  // rolling 9-base window
      for (int64_t i = 0; i < L; ++i) {
        const unsigned b = seq[i];
        digits = ((digits << 2) | (b ? b - 1u : 0u)) & 0x3FFFFu;
        invalid = ((invalid << 1) | (b == 0)) & 0x1FFu;
        for (int k = 0; k <= 2 && k <= i; ++k) {
          const unsigned y = digits & ((1u << (2 * (k + 1))) - 1u);
          if (invalid == 0 || y == 0) ++d[k][y];
        }
      }
    }
    }
  };
  {
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
  }
  for (unsigned t = 0; t < nt; ++t) {
    const long long* f = part[t].data();
    const long long* e = part[t].data() + 84;
    // fast path: f[0..4) first bases, f[4..20) first 2-mers, f[20..84) 3-mers ending at i >= 2
    long long n2[16], n1[4];
    for (int ab = 0; ab < 16; ++ab) {
      n2[ab] = f[4 + ab];
      for (int x = 0; x < 4; ++x) n2[ab] += f[20 + x * 16 + ab];
    }
    for (int a = 0; a < 4; ++a) {
      n1[a] = f[a];
      for (int x = 0; x < 4; ++x) n1[a] += n2[x * 4 + a];
    }
    for (int y = 0; y < 4; ++y) n_[0][y] += n1[y] + e[y];
    for (int y = 0; y < 16; ++y) n_[1][y] += n2[y] + e[4 + y];
    for (int y = 0; y < 64; ++y) n_[2][y] += f[20 + y] + e[20 + y];
  }
  // sharded ingest: the model is additive in its 84 counters (not in V)
  for (int k = 0; k < 3; ++k) SequenceSet::allreduceSum(n_[k], (size_t)1 << (2 * (k + 1)));
  calculateV();
}

BackgroundModel::~BackgroundModel() {
  for (int k = 0; k < 3; ++k) {
    delete[] n_[k];
    delete[] v_[k];
  }
}

// The reference keeps the counters and their sum in `int` (src/shared/BackgroundModel.cpp:492-495), which overflows
// beyond 2^31 bases (BASELINE configs[3]: 2e10).  Here they are 64-bit and converted to float directly: bit-identical
// to the reference while every counter fits an int, the intended arithmetic beyond (documented like the 64-bit scan
// positions of the count; the device's pengk_bg_model does the same).
void BackgroundModel::calculateV() {
  long long base_counts = 0;
  for (int y = 0; y < 4; ++y) base_counts += n_[0][y];
  for (int y = 0; y < 4; ++y) v_[0][y] = ((float)n_[0][y] + A_[0] * 0.25f) / ((float)base_counts + A_[0]);
  for (int k = 1; k <= K_; ++k) {
    const int ny = 1 << (2 * (k + 1)), yk = 1 << (2 * k);
    for (int y = 0; y < ny; ++y) {
      const float prior = interpolate_ ? v_[k - 1][y % yk] : 0.25f;
      v_[k][y] = ((float)n_[k][y] + A_[k] * prior) / ((float)n_[k - 1][y / 4] + A_[k]);
    }
    for (int g = 0; g < ny; g += 4) {
      float factor = 0.0f;
      for (int a = 0; a < 4; ++a) factor += v_[k][g + a];
      for (int a = 0; a < 4; ++a) v_[k][g + a] /= factor;
    }
  }
}
