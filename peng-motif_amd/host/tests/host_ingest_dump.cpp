// host_ingest_dump -- CPU-only probe of the host ingest (SequenceSet + BackgroundModel): prints what the
// parity test compares with the oracle.  usage: host_ingest_dump FASTA
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "shared/BackgroundModel.h"
#include "shared/SequenceSet.h"

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  Alphabet::init("STANDARD");
  SequenceSet ss(argv[1], true);
  BackgroundModel bg(ss, 2, std::vector<float>{1.f, 1.f, 1.f}, true);
  const size_t N = ss.getN();
  uint64_t h = 1469598103934665603ull;  // FNV-1a over the codes
  for (int64_t i = 0; i < ss.offsets()[N]; ++i) h = (h ^ ss.codes()[i]) * 1099511628211ull;
  std::printf("N %zu minL %u maxL %u total %lld fnv %016llx\n", N, ss.getMinL(), ss.getMaxL(), (long long)ss.offsets()[N],
              (unsigned long long)h);
  std::printf("counts");
  for (int k = 0; k < 3; ++k)
    for (int y = 0; y < (1 << (2 * (k + 1))); ++y) std::printf(" %lld", bg.getCounts(k)[y]);
  std::printf("\nV");
  for (int k = 0; k < 3; ++k)
    for (int y = 0; y < (1 << (2 * (k + 1))); ++y) {
      uint32_t u;
      std::memcpy(&u, &bg.getV()[k][y], 4);
      std::printf(" %08x", u);
    }
  std::printf("\n");
  // views materialise lazily and agree with the contiguous buffer
  std::vector<Sequence*> seqs = ss.getSequences();
  bool ok = seqs.size() == N;
  for (size_t i = 0; ok && i < N; i += (N / 50 + 1))
    ok = seqs[i]->getL() == (int)(ss.offsets()[i + 1] - ss.offsets()[i]) && seqs[i]->getSequence() == ss.codes() + ss.offsets()[i];
  std::printf("views %s\n", ok ? "ok" : "BAD");
  if (N) std::printf("header0 %s\n", seqs[0]->getHeader().c_str());
  return ok ? 0 : 1;
}
