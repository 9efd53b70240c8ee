// ASan/UBSan harness for the host packer (pack.cpp) -- build container only
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>
#include "pengk_internal.h"
namespace pengk { int fail(int code, const char* fmt, ...) { (void)fmt; return code; } int hip_fail(hipError_t, const char*) { return PENGK_ERR_DEVICE; } }
int main() {
  std::mt19937_64 rng(7);
  int cases = 0;
  for (int W : {4, 8, 10, 14}) for (int M : {64, 256}) for (int mode = 0; mode < 4; ++mode) for (const char* nt : {"1", "3", "16"}) {
    setenv("PENGK_PACK_THREADS", nt, 1);
    std::vector<int64_t> offs(1, 0); std::vector<uint8_t> codes;
    int n = 700;
    for (int s = 0; s < n; ++s) {
      int L = mode == 0 ? W + (int)(rng() % 400) : (int)(rng() % 300);
      if (mode == 3 && s % 50 == 0) L = 3000 + (int)(rng() % 2000);
      for (int i = 0; i < L; ++i) { unsigned c = 1 + (unsigned)(rng() % 4); if (mode >= 2 && rng() % 97 == 0) c = 0; codes.push_back((uint8_t)c); }
      offs.push_back((int64_t)codes.size());
    }
    pengk_packed pk;
    int rc = pengk_pack(codes.data(), offs.data(), n, W, M, &pk);
    if (rc) { printf("rc %d\n", rc); return 1; }
    pengk_packed_free(&pk);
    ++cases;
  }
  // chunks of an input packed into ONE pair of buffers from several threads (the CLI's streaming ingest)
  for (int W : {6, 10, 14}) for (int mode = 0; mode < 3; ++mode) {
    const int n_chunks = 9, n = 300;
    std::vector<std::vector<uint8_t>> cc(n_chunks);
    std::vector<std::vector<int64_t>> oo(n_chunks, std::vector<int64_t>(1, 0));
    size_t bases = 0, recs = 0;
    for (int k = 0; k < n_chunks; ++k)
      for (int s = 0; s < n; ++s) {
        int L = mode == 0 ? W + (int)(rng() % 200) : (int)(rng() % 250);
        for (int i = 0; i < L; ++i) { unsigned c = 1 + (unsigned)(rng() % 4); if (mode == 2 && rng() % 61 == 0) c = 0; cc[k].push_back((uint8_t)c); }
        oo[k].push_back((int64_t)cc[k].size());
        bases += (size_t)L; ++recs;
      }
    std::vector<uint64_t> words(bases / 32 + n_chunks * 8 + 16, 0), items(bases / (W + 1) + bases / 256 + n_chunks + recs + 16, 0);
    pengk_pack_target tg = {words.data(), words.size(), items.data(), items.size(), 0, 0};
    std::vector<std::thread> th;
    std::vector<int> rcs(n_chunks, -1);
    for (int k = 0; k < n_chunks; ++k)
      th.emplace_back([&, k] { pengk_packed part; rcs[k] = pengk_pack_append(cc[k].data(), oo[k].data(), n, W, 0, &tg, &part); });
    for (auto& t : th) t.join();
    for (int rc : rcs) if (rc) { printf("append rc %d\n", rc); return 1; }
    if (tg.word_cursor > words.size() || tg.item_cursor > items.size()) return 1;
    ++cases;
  }
  // empty input, single short sequence
  pengk_packed pk; int64_t o[2] = {0, 3}; uint8_t c[3] = {1, 2, 3};
  if (pengk_pack(c, o, 1, 8, 0, &pk)) return 1; pengk_packed_free(&pk);
  if (pengk_pack(nullptr, nullptr, 0, 8, 0, &pk)) return 1; pengk_packed_free(&pk);
  printf("ok %d\n", cases);
}
