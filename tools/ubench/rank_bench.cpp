#include <chrono>
#include <cstdio>
#include <random>
#include <vector>
#include "ranked_prefix.h"
int main(int argc, char** argv) {
  for (size_t n : {(size_t)1 << 20, (size_t)1 << 24}) {
    std::mt19937_64 rng(1);
    std::normal_distribution<float> nd(0.f, 2.f);
    std::vector<float> z(n);
    for (auto& x : z) x = nd(rng);
    for (size_t i = 0; i < n / 2; ++i) z[n - 1 - i] = z[i];  // twin ties
    for (int i = 0; i < 300; ++i) z[rng() % n] = 12.f + (float)(i % 50);
    ranked_prefix::EntryVec e;
    for (int rep = 0; rep < 3; ++rep) {
      auto t0 = std::chrono::steady_clock::now();
      size_t m = ranked_prefix::rank(z.data(), n, 10.f, e);
      double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      printf("n=%zu ranked %zu  %.2f ms\n", n, m, ms);
    }
  }
}
