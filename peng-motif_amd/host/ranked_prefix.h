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
#include <cstdlib>
#include <memory>
#include <new>
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

typedef std::vector<Entry> EntryVec;  // (transparent huge pages for it were tried: no gain on the GPU box's host, and the
// threaded partitions ran five times SLOWER on them in the build container; profiles/r04_rank_bench.log)

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

// The same partition, by several threads, for the large ranges at the top of the replay (the first three or four
// partitions are ~90 % of its work: 16.7 M entries at W = 12).  The scalar loop above pairs the k-th "left stopper" --
// the k-th element from the left that is NOT before the pivot -- with the k-th "right stopper" -- the k-th element from
// the right that the pivot is NOT before --, swaps them and goes on until the two meet; a swapped element is never looked
// at again (both scans have passed it), so the pairs are those of the ORIGINAL array: stoppers can be listed first, by
// chunks, then the crossing K = the first k with L_k >= R_k is found (L ascends, R descends), the K pairs are swapped in
// any order, and the result -- the arrangement, and the return value (where the left scan stops last: L_K, or R_(K-1) if
// that comes first, see below) -- is the scalar loop's, element for element
// (host/tests/ranked_prefix_test.cpp runs both against std::sort on tie-heavy arrays of up to 2^22 entries).
inline Entry* partition_parallel(Entry* first, Entry* last, const Entry* pivot, unsigned nt) {
  const size_t n = (size_t)(last - first);
  const float pz = pivot->z;
  // stopper lists per chunk, in uninitialised storage (a vector would zero-fill what is overwritten at once): ONE block
  // for all of them, 8 bytes per entry of the range, allocated here -- a thread body must not throw -- and if it is not
  // to be had the scalar loop does the work (same arrangement, same return value)
  struct List {
    uint32_t* p = nullptr;
    size_t n = 0;
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    uint32_t operator[](size_t i) const { return p[i]; }
  };
  std::unique_ptr<uint32_t[]> block(new (std::nothrow) uint32_t[2 * (n + nt)]);
  if (!block) return partition(first, last, pivot);
  std::vector<List> Ls(nt), Rs(nt);
  {
    auto scan = [&](unsigned t) {
      const size_t lo = n * t / nt, hi = n * (t + 1) / nt;
      uint32_t* l = block.get() + 2 * (lo + t);  // (chunk t's two lists of hi - lo + 1 entries: the chunks in front hold 2 (lo + t))
      uint32_t* r = l + (hi - lo + 1);
      size_t nl = 0, nr = 0;
      for (size_t i = lo; i < hi; ++i) {  // (branch-free appends: a random table mispredicts every other compare)
        const float zi = first[i].z;
        l[nl] = (uint32_t)i;
        nl += !(zi > pz);
        r[nr] = (uint32_t)i;
        nr += !(pz > zi);
      }
      Ls[t].p = l;
      Ls[t].n = nl;
      Rs[t].p = r;
      Rs[t].n = nr;
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(scan, t);
    scan(0);
    for (auto& x : th) x.join();
  }
  std::vector<size_t> preL(nt + 1, 0), preR(nt + 1, 0);  // stoppers in the chunks in front of chunk t
  for (unsigned t = 0; t < nt; ++t) {
    preL[t + 1] = preL[t] + Ls[t].size();
    preR[t + 1] = preR[t] + Rs[t].size();
  }
  const size_t totL = preL[nt], totR = preR[nt];
  auto Lk = [&](size_t k) {  // position of the k-th left stopper
    const unsigned t = (unsigned)(std::upper_bound(preL.begin(), preL.end(), k) - preL.begin()) - 1u;
    return (size_t)Ls[t][k - preL[t]];
  };
  auto Rk = [&](size_t k) {  // position of the k-th right stopper counted from the right
    const size_t a = totR - 1 - k;  // its ascending rank
    const unsigned t = (unsigned)(std::upper_bound(preR.begin(), preR.end(), a) - preR.begin()) - 1u;
    return (size_t)Rs[t][a - preR[t]];
  };
  // K = the first k with L_k >= R_k.  (The median-of-three pivot guarantees the scalar loop a stopper on either side
  // before it leaves the range; should the lists end first, the range is handed to the scalar loop untouched.)
  const size_t kmax = std::min(totL, totR);
  size_t lo = 0, hi = kmax;
  while (lo < hi) {
    const size_t mid = lo + (hi - lo) / 2;
    if (Lk(mid) >= Rk(mid)) hi = mid;
    else lo = mid + 1;
  }
  const size_t K = lo;  // (at k = kmax one of the lists has ended: the pointers have met by then at the latest)
  // Where the left scan stops in its last round: at L_K -- or earlier, at R_(K-1), which now holds what the last swap
  // brought over from the left, a left stopper by definition (in the ORIGINAL array that place need not be one).
  if (K == 0 && totL == 0) return partition(first, last, pivot);  // (cannot happen behind a median-of-three pivot)
  size_t cut = K < totL ? Lk(K) : n;
  if (K > 0) cut = std::min(cut, Rk(K - 1));
  {
    auto swaps = [&](unsigned t) {
      size_t k = K * t / nt;
      const size_t k1 = K * (t + 1) / nt;
      if (k >= k1) return;
      unsigned tl = (unsigned)(std::upper_bound(preL.begin(), preL.end(), k) - preL.begin()) - 1u;
      size_t il = k - preL[tl];
      size_t a = totR - 1 - k;
      unsigned tr = (unsigned)(std::upper_bound(preR.begin(), preR.end(), a) - preR.begin()) - 1u;
      size_t ir = a - preR[tr];
      for (; k < k1; ++k) {
        std::swap(first[Ls[tl][il]], first[Rs[tr][ir]]);
        if (++il == Ls[tl].size()) {
          il = 0;
          do ++tl;
          while (tl < nt && Ls[tl].empty());
        }
        if (ir == 0) {
          do --tr;
          while (tr != (unsigned)-1 && Rs[tr].empty());
          ir = tr != (unsigned)-1 ? Rs[tr].size() - 1 : 0;
        } else {
          --ir;
        }
      }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(swaps, t);
    swaps(0);
    for (auto& x : th) x.join();
  }
  return first + cut;
}

// Ranges from 2^18 entries on are partitioned by several threads, one per 2^17 entries and 16 at most (on the GPU box's
// host 1 M entries rank in 7.9 ms on one thread, 4.6 on 8, 7.2 on 16, 14 on 32; 16.7 M in 71 / 22 / 17 / 16 ms:
// profiles/r04_rank_bench.log).  PENGK_RANK_THREADS caps the number (1 = the scalar loop everywhere).
inline unsigned partition_threads(size_t n) {
  static const unsigned cap = [] {
    const char* e = std::getenv("PENGK_RANK_THREADS");
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    return e && *e ? (unsigned)std::max(1, std::atoi(e)) : std::min(16u, hw);
  }();
  // (above 2^27 entries -- the first partition of a W = 14 table -- the stopper lists would take more than a GiB beside the
  // table: that one range goes through the scalar loop, its halves through the threads)
  if (n < ((size_t)1 << 18) || n > ((size_t)1 << 27)) return 1u;
  return (unsigned)std::min<size_t>(cap, n >> 17);
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
    const unsigned nt = partition_threads((size_t)(last - first));
    Entry* cut = nt > 1 ? partition_parallel(first + 1, last, first, nt) : partition(first + 1, last, first);
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
inline size_t rank(const float* z, size_t n, float threshold, EntryVec& entries) {
  entries.resize(n);
  bool ordered = true;  // a NaN breaks the "right of the pivot is smaller" argument: sort everything then
  {
    // 4^12 entries are 134 MB: filled by several threads (the replay itself is sequential by nature)
    unsigned nt = n < ((size_t)1 << 18) ? 1u : std::min(n < ((size_t)1 << 22) ? 4u : 16u, std::max(1u, std::thread::hardware_concurrency()));
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
