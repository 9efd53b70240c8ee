// ranked_prefix.h -- the part of `std::sort(order, order + 4^W, sort_indices(z))` that seed selection reads.
//
// BasePattern::select_base_patterns (reference src/base_pattern.cpp:443-515) sorts ALL 4^W pattern ids by
// z-score with a non-stable std::sort and then walks the ranking only until the first z < threshold
// (:463).  Which of two ids with exactly equal z (every reverse-complement pair in BOTH mode) comes
// first decides which strand a seed is reported on, so the ranking has to be the one std::sort leaves,
// not merely "a" descending order.  This header replays libstdc++'s introsort step by step -- same
// median-of-three pivots, same unguarded partition, same depth limit with the heap-sort fallback, same
// final insertion sort -- on (z, id) pairs, but does not sort a right-hand partition whose pivot is
// already below the threshold: everything in it and to the right of it is < threshold and is never
// read by the walk.  The returned prefix [0, n_ranked) is element for element what std::sort produces;
// 4^W·log(4^W) comparisons become ~2·4^W.  (host/tests/ranked_prefix_test.cpp compares the two on this
// machine's libstdc++, ties and all.)
#ifndef PENGK_HOST_RANKED_PREFIX_H_
#define PENGK_HOST_RANKED_PREFIX_H_

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <thread>
#include <utility>
#include <vector>

namespace ranked_prefix {

struct Entry {
  float z;
  uint32_t id;
  Entry() {}  // (no initialisation: vector::resize must not zero-fill 4^W entries that are overwritten at once)
  Entry(float z_, uint32_t id_) : z(z_), id(id_) {}
};

// the comparator of sort_indices (src/base_pattern.h:166-172): descending z
inline bool before(const Entry& a, const Entry& b) { return a.z > b.z; }

namespace detail {

constexpr std::ptrdiff_t kChunk = 16;  // below this size introsort leaves a range to the final insertion sort

inline void median_to_first(Entry* result, Entry* a, Entry* b, Entry* c) {
  if (before(*a, *b)) {
    if (before(*b, *c))
      std::swap(*result, *b);
    else if (before(*a, *c))
      std::swap(*result, *c);
    else
      std::swap(*result, *a);
  } else if (before(*a, *c)) {
    std::swap(*result, *a);
  } else if (before(*b, *c)) {
    std::swap(*result, *c);
  } else {
    std::swap(*result, *b);
  }
}

inline Entry* partition(Entry* first, Entry* last, const Entry* pivot) {
  for (;;) {
    while (before(*first, *pivot)) ++first;
    --last;
    while (before(*pivot, *last)) --last;
    if (!(first < last)) return first;
    std::swap(*first, *last);
    ++first;
  }
}

inline void quick_loop(Entry* first, Entry* last, long depth, float threshold, Entry*& ranked_end) {
  while (last - first > kChunk) {
    if (depth == 0) {
      std::partial_sort(first, last, last, before);  // the heap sort std::sort falls back to
      return;
    }
    --depth;
    Entry* mid = first + (last - first) / 2;
    median_to_first(first, first + 1, mid, last - 1);
    Entry* cut = partition(first + 1, last, first);
    if (first->z < threshold) {
      // [cut, last) holds only z <= pivot < threshold, and so does everything to its right: never read
      if (cut < ranked_end) ranked_end = cut;
    } else {
      quick_loop(cut, last, depth, threshold, ranked_end);
    }
    last = cut;
  }
}

inline void linear_insert(Entry* last) {
  const Entry val = *last;
  Entry* next = last - 1;
  while (before(val, *next)) {
    *last = *next;
    last = next;
    --next;
  }
  *last = val;
}

inline void insertion(Entry* first, Entry* last) {
  if (first == last) return;
  for (Entry* i = first + 1; i != last; ++i) {
    if (before(*i, *first)) {
      const Entry val = *i;
      std::move_backward(first, i, i + 1);
      *first = val;
    } else {
      linear_insert(i);
    }
  }
}

}  // namespace detail

// Ranks ids 0..n-1 by z.  On return entries[0 .. n_ranked) equal the first n_ranked elements of
// std::sort(ids, sort_indices(z)); n_ranked == n, or every id ranked at or after n_ranked has z < threshold.
inline size_t rank(const float* z, size_t n, float threshold, std::vector<Entry>& entries) {
  entries.resize(n);
  bool ordered = true;  // a NaN breaks the "right of the pivot is smaller" argument: sort everything then
  {
    // 4^12 entries are 134 MB: filled by several threads (the replay itself is sequential by nature)
    unsigned nt = n < ((size_t)1 << 22) ? 1u : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    std::vector<char> nan(nt, 0);
    Entry* e = entries.data();
    auto fill = [&](unsigned t) {
      bool any = false;
      for (size_t i = n * t / nt; i < n * (t + 1) / nt; ++i) {
        e[i] = Entry(z[i], (uint32_t)i);
        any |= z[i] != z[i];
      }
      nan[t] = any;
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(fill, t);
    fill(0);
    for (auto& x : th) x.join();
    for (char c : nan) ordered &= !c;
  }
  if (!ordered || threshold != threshold) threshold = -INFINITY;
  if (n == 0) return 0;
  Entry* first = entries.data();
  Entry* ranked_end = first + n;
  long lg = 0;
  for (size_t m = n; m > 1; m >>= 1) ++lg;
  detail::quick_loop(first, first + n, 2 * lg, threshold, ranked_end);
  const std::ptrdiff_t m = ranked_end - first;
  if ((std::ptrdiff_t)n > detail::kChunk && m > detail::kChunk) {
    detail::insertion(first, first + detail::kChunk);
    for (Entry* i = first + detail::kChunk; i != ranked_end; ++i) detail::linear_insert(i);
  } else {
    detail::insertion(first, ranked_end);
  }
  return (size_t)m;
}

}  // namespace ranked_prefix
#endif
