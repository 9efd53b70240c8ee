// ranked_prefix_test.cpp -- CPU-only: the pruned replay of std::sort in ranked_prefix.h against std::sort
// itself (same comparator as the reference's sort_indices), on arrays full of exact ties.
// Prints "ok <cases>" and exits 0, or the first mismatch and exits 1.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "ranked_prefix.h"

namespace {
struct by_z {
  const float* z;
  bool operator()(size_t a, size_t b) const { return z[a] > z[b]; }
};

int check(const std::vector<float>& z, float thr, const char* what) {
  const size_t n = z.size();
  std::vector<size_t> order(n);
  for (size_t i = 0; i < n; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), by_z{z.data()});
  ranked_prefix::EntryVec e;
  const size_t m = ranked_prefix::rank(z.data(), n, thr, e);
  if (m > n) {
    printf("FAIL %s: prefix %zu > n %zu\n", what, m, n);
    return 1;
  }
  for (size_t r = 0; r < m; ++r)
    if (e[r].id != order[r] || !(e[r].z == z[order[r]] || (e[r].z != e[r].z))) {
      printf("FAIL %s (n=%zu thr=%g): rank %zu is id %u, std::sort has %zu\n", what, n, thr, r, e[r].id, order[r]);
      return 1;
    }
  if (m < n) {
    // everything that was not ranked must be below the threshold, and so must be std::sort's element at m
    for (size_t r = m; r < n; ++r)
      if (!(z[order[r]] < thr)) {
        printf("FAIL %s (n=%zu thr=%g): unranked id %zu has z %g\n", what, n, thr, order[r], z[order[r]]);
        return 1;
      }
  }
  return 0;
}
}  // namespace

int main() {
  std::mt19937_64 rng(12345);
  int cases = 0;
  // (from 2^18 entries on the large partitions run on several threads: ranked_prefix.h, partition_parallel)
  const size_t sizes[] = {0, 1, 2, 15, 16, 17, 18, 33, 100, 1000, 4096, 65536, 131071, 131072, 131073, 300001, 1u << 20, (1u << 22) + 5};
  for (size_t n : sizes) {
    for (int style = 0; style < 6; ++style) {
      std::vector<float> z(n);
      for (size_t i = 0; i < n; ++i) {
        switch (style) {
          case 0: z[i] = (float)(int)(rng() % 50) - 20.f; break;                        // very few distinct values
          case 1: z[i] = (float)(int)(rng() % 5000) / 7.f - 100.f; break;               // many ties
          case 2: z[i] = std::ldexp((float)(rng() % (1u << 24)), -18) - 30.f; break;    // nearly distinct
          case 3: z[i] = (float)i; break;                                               // ascending (worst for naive pivots)
          case 4: z[i] = -(float)(i / 3); break;                                        // descending with ties
          default: z[i] = (i % 2) ? 3.f : (float)(rng() % 7); break;
        }
      }
      if (style == 1)  // mirror pairs like a BOTH table: z[i] == z[n-1-i]
        for (size_t i = 0; i < n / 2; ++i) z[n - 1 - i] = z[i];
      const float thrs[] = {-INFINITY, -25.f, 0.f, 3.f, 10.f, 600.f, INFINITY};
      for (float thr : thrs) {
        if (check(z, thr, "random")) return 1;
        ++cases;
      }
    }
  }
  {  // +inf scores (expected count 0) and a NaN: the NaN case must fall back to the full sort
    std::vector<float> z(5000);
    for (size_t i = 0; i < z.size(); ++i) z[i] = (float)(int)(rng() % 100);
    z[17] = INFINITY;
    z[4000] = INFINITY;
    if (check(z, 10.f, "inf")) return 1;
    z[123] = NAN;
    ranked_prefix::EntryVec e;
    if (ranked_prefix::rank(z.data(), z.size(), 10.f, e) != z.size()) {
      printf("FAIL nan: prefix must be the whole array\n");
      return 1;
    }
    cases += 2;
  }
  printf("ok %d\n", cases);
  return 0;
}
