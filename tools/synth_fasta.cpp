// synth_fasta -- the synthetic FASTA of SURVEY.md 8(d) (counter-based splitmix64 generator, motif GCTGAGTCAT planted
// in 10 % of the sequences), written with all host threads.  Same bits as pengk_synth_sequences and the oracle's
// po_synth; used by bench.py to time the peng_motif CLI end to end on BASELINE configs[2].
//   synth_fasta OUT.fa N_SEQ [L=200] [SEED=1] [SEQ0=0]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

static inline uint64_t mix64(uint64_t x) {
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: synth_fasta OUT.fa N_SEQ [L=200] [SEED=1] [SEQ0=0]\n");
    return 2;
  }
  const uint64_t n = strtoull(argv[2], nullptr, 10);
  const uint32_t L = argc > 3 ? (uint32_t)atoi(argv[3]) : 200u;
  const uint64_t seed = argc > 4 ? strtoull(argv[4], nullptr, 10) : 1ull;
  const uint64_t seq0 = argc > 5 ? strtoull(argv[5], nullptr, 10) : 0ull;
  unsigned nt = std::thread::hardware_concurrency();
  if (nt == 0) nt = 1;
  if (nt > 64) nt = 64;
  if (n < nt) nt = 1;
  std::vector<std::string> part(nt);
  std::vector<std::thread> th;
  auto work = [&](unsigned t) {
    const uint64_t lo = n * t / nt, hi = n * (t + 1) / nt;
    std::string& out = part[t];
    out.reserve((size_t)(hi - lo) * (L + 14));
    static const char motif[] = "GCTGAGTCAT";
    char head[32];
    for (uint64_t i = lo; i < hi; ++i) {
      const uint64_t s = seq0 + i;
      out.append(head, (size_t)snprintf(head, sizeof head, ">s%llu\n", (unsigned long long)s));
      const size_t at = out.size();
      out.resize(at + L + 1);
      char* row = &out[at];
      for (uint32_t j = 0; j < L; ++j) row[j] = "ACGT"[mix64(seed + 0x9E3779B97F4A7C15ull * (s * (uint64_t)L + j + 1)) >> 62];
      if (L >= 10 && mix64(seed ^ 0xA5A5A5A5ull ^ (s + 1)) % 10 == 0) {
        const uint64_t q = mix64(seed ^ 0x5A5A5A5Aull ^ (s + 1)) % (L - 9);
        for (int k = 0; k < 10; ++k) row[q + k] = motif[k];
      }
      row[L] = '\n';
    }
  };
  for (unsigned t = 1; t < nt; ++t) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
  FILE* f = fopen(argv[1], "wb");
  if (!f) {
    perror(argv[1]);
    return 1;
  }
  for (auto& p : part)
    if (fwrite(p.data(), 1, p.size(), f) != p.size()) {
      perror("write");
      return 1;
    }
  fclose(f);
  return 0;
}
