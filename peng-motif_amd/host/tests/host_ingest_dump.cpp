// host_ingest_dump -- CPU-only probe of the host ingest (SequenceSet + BackgroundModel): prints what the
// parity test compares with the oracle.
//   host_ingest_dump FASTA                         the whole file, one process
//   host_ingest_dump FASTA RANK WORLD DIR [CODES]  sharded ingest: this process is rank RANK of WORLD; the ranks combine
//                                                  their shards through files in DIR (a stand-in for the host channel
//                                                  of include/pengk.h); CODES receives this rank's code bytes
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "shared/BackgroundModel.h"
#include "shared/SequenceSet.h"

namespace {
int g_rank = 0, g_world = 1, g_round = 0;
std::string g_dir;

// recv[r * bytes ...) = rank r's send: every rank writes a file per round and waits for the others' (up to 60 s)
bool file_allgather(const void* send, void* recv, size_t bytes) {
  const int round = g_round++;
  auto name = [&](int r) { return g_dir + "/gather_" + std::to_string(round) + "_" + std::to_string(r); };
  const std::string tmp = name(g_rank) + ".tmp";
  FILE* f = std::fopen(tmp.c_str(), "wb");
  if (!f) return false;
  const bool wrote = std::fwrite(send, 1, bytes, f) == bytes;
  std::fclose(f);
  if (!wrote || std::rename(tmp.c_str(), name(g_rank).c_str()) != 0) return false;
  for (int r = 0; r < g_world; ++r) {
    FILE* in = nullptr;
    for (int spin = 0; spin < 6000 && !(in = std::fopen(name(r).c_str(), "rb")); ++spin) usleep(10000);
    if (!in) return false;
    const bool read = std::fread((char*)recv + (size_t)r * bytes, 1, bytes, in) == bytes;
    std::fclose(in);
    if (!read) return false;
  }
  return true;
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  if (argc >= 5) {
    g_rank = std::atoi(argv[2]);
    g_world = std::atoi(argv[3]);
    g_dir = argv[4];
    SequenceShardComm sc;
    sc.rank = g_rank;
    sc.world = g_world;
    sc.allgather = file_allgather;
    SequenceSet::setShardComm(sc);
  }
  Alphabet::init("STANDARD");
  SequenceSet ss(argv[1], true);
  BackgroundModel bg(ss, 2, std::vector<float>{1.f, 1.f, 1.f}, true);
  const size_t N = ss.getLocalN();
  uint64_t h = 1469598103934665603ull;  // FNV-1a over the codes
  for (int64_t i = 0; i < ss.offsets()[N]; ++i) h = (h ^ ss.codes()[i]) * 1099511628211ull;
  std::printf("N %zu minL %u maxL %u total %lld fnv %016llx localN %zu base %zu", ss.getN(), ss.getMinL(), ss.getMaxL(),
              (long long)ss.offsets()[N], (unsigned long long)h, N, ss.getLocalBase());
  for (int i = 0; i < 4; ++i) {
    uint32_t u;
    std::memcpy(&u, &ss.getBaseFrequencies()[i], 4);
    std::printf(" f%d %08x", i, u);
  }
  std::printf("\ncounts");
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
  // views materialise lazily (they point into the reader's chunks) and agree with the contiguous copy
  std::vector<Sequence*> seqs = ss.getSequences();
  bool ok = seqs.size() == N;
  for (size_t i = 0; ok && i < N; i += (N / 50 + 1))
    ok = seqs[i]->getL() == (int)(ss.offsets()[i + 1] - ss.offsets()[i]) &&
         std::memcmp(seqs[i]->getSequence(), ss.codes() + ss.offsets()[i], (size_t)seqs[i]->getL()) == 0;
  size_t in_chunks = 0;
  for (size_t c = 0; c < ss.nChunks(); ++c) {
    ok = ok && ss.chunk(c).first == in_chunks;
    in_chunks += ss.chunk(c).n;
  }
  ok = ok && in_chunks == N;
  std::printf("views %s\n", ok ? "ok" : "BAD");
  if (N) std::printf("header0 %s\n", seqs[0]->getHeader().c_str());
  if (argc >= 6) {  // this rank's codes and record lengths, for the test to lay end to end
    FILE* f = std::fopen(argv[5], "wb");
    if (!f) return 1;
    std::fwrite(ss.codes(), 1, (size_t)ss.offsets()[N], f);
    std::fclose(f);
    f = std::fopen((std::string(argv[5]) + ".offs").c_str(), "wb");
    if (!f) return 1;
    std::fwrite(ss.offsets(), sizeof(int64_t), N + 1, f);
    std::fclose(f);
  }
  return ok ? 0 : 1;
}
