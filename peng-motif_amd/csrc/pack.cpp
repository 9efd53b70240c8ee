// pack.cpp -- host packer: reference byte codes -> 2-bit stream + scan items + background counts.
//
// Replaces what BasePattern::count_patterns reads per sequence (byte codes, a freshly built
// reverse-complement copy and the scan bookkeeping, src/base_pattern.cpp:339-381) and what
// Sequence/BackgroundModel precompute for the background model (src/shared/Sequence.cpp:28-33,
// src/shared/BackgroundModel.cpp:60-84).  Pure CPU, no HIP calls: usable without a GPU.
//
// The scan rule of src/base_pattern.cpp:347-381 is resolved HERE, once: a sequence is cut into
// "visited runs" (>= W valid bases).  A run that is ended by an invalid base at q resumes the scan
// at q+2 (the base at q+1 is never part of a window); a failed build (fewer than W valid bases
// before an invalid base at q) resumes at q+1.  Only run bases are stored, so the device never
// sees an invalid base; windows of different runs are >= W+1 apart and cannot suppress each other
// under the non-overlap rule (:361-366), so runs are independent work items.
//
// Two passes over the sequences, both split over host threads by contiguous sequence ranges:
// (1) measure -- runs, windows, items, background counters per range; (2) write -- every range knows its
// offsets in the stream and the item table from a prefix sum and fills them independently (words shared
// by two ranges are merged with atomic OR).  Output is identical for any thread count.
#include <stdlib.h>
#include <string.h>

#include <new>
#include <thread>
#include <vector>

#include "pengk_internal.h"

using namespace pengk;

namespace {

struct RangeStat {
  uint64_t bases = 0, windows = 0, items = 0, bound = 0, max_len = 0;
  int64_t bg[84] = {0};
  int all_whole = 1;
};

// f(run_start, run_len) for every visited run of one sequence
template <class F>
inline void for_each_run(const uint8_t* seq, int64_t L, int W, F&& f) {
  int64_t i = 0;
  while (i < L) {
    int64_t j = i;
    while (j < L && (uint8_t)(seq[j] - 1) < 4) ++j;
    if (j - i >= W) {
      f(i, j - i);
      i = j + 2;
    } else {
      i = j + 1;
    }
  }
}

void measure_range(const uint8_t* codes, const int64_t* offs, int64_t s0, int64_t s1, int W, uint64_t M, RangeStat& st) {
  int64_t* bgk[3] = {st.bg, st.bg + 4, st.bg + 20};
  for (int64_t s = s0; s < s1; ++s) {
    const uint8_t* seq = codes + offs[s];
    const int64_t L = offs[s + 1] - offs[s];
    if ((uint64_t)L > st.max_len) st.max_len = (uint64_t)L;
    // background (k+1)-mer counts with the reference's invalid-base behaviour: with an invalid base among
    // the last (up to) 9 positions only an all-zero (k+1)-mer (invalid = digit 0) is counted
    uint32_t d9 = 0, m9 = 0;
    for (int64_t i = 0; i < L; ++i) {
      const unsigned c = seq[i];
      const unsigned inv = (c == 0 || c > 4);
      d9 = ((d9 << 2) | (inv ? 0u : c - 1u)) & 0x3FFFFu;
      m9 = ((m9 << 1) | inv) & 0x1FFu;
      if (m9 == 0) {
        ++bgk[0][d9 & 3u];
        if (i >= 1) ++bgk[1][d9 & 15u];
        if (i >= 2) ++bgk[2][d9 & 63u];
      } else {
        if ((d9 & 3u) == 0) ++bgk[0][0];
        if (i >= 1 && (d9 & 15u) == 0) ++bgk[1][0];
        if (i >= 2 && (d9 & 63u) == 0) ++bgk[2][0];
      }
    }
    int runs = 0;
    bool whole = false;
    for_each_run(seq, L, W, [&](int64_t start, int64_t len) {
      ++runs;
      whole = (start == 0 && len == L);
      const uint64_t nwin = (uint64_t)(len - W + 1);
      st.bases += (uint64_t)len;
      st.windows += nwin;
      st.bound += (nwin + W - 1) / W;
      st.items += (nwin + M - 1) / M;
    });
    if (!(runs == 1 && whole)) st.all_whole = 0;
  }
}

void write_range(const uint8_t* codes, const int64_t* offs, int64_t s0, int64_t s1, int W, uint64_t M, uint64_t base0,
                 uint64_t item0, uint64_t* words, uint64_t* items) {
  uint64_t g = base0;  // next stream position
  uint64_t acc = 0;    // bits gathered for word g >> 5 (only the bits this range owns)
  uint64_t it = item0;
  auto flush = [&](uint64_t word) {
    if (acc) __atomic_fetch_or(&words[word], acc, __ATOMIC_RELAXED);
    acc = 0;
  };
  for (int64_t s = s0; s < s1; ++s) {
    const uint8_t* seq = codes + offs[s];
    const int64_t L = offs[s + 1] - offs[s];
    for_each_run(seq, L, W, [&](int64_t start, int64_t len) {
      const uint64_t run0 = g;
      for (int64_t t = start; t < start + len; ++t) {
        acc |= (uint64_t)(seq[t] - 1u) << (2 * (g & 31));
        ++g;
        if ((g & 31) == 0) flush((g >> 5) - 1);
      }
      const uint64_t nwin = (uint64_t)(len - W + 1);
      for (uint64_t f = 0; f < nwin; f += M) {
        const uint64_t nw = nwin - f < M ? nwin - f : M;
        items[it++] = (run0 + f) | (nw << ITEM_NW_SHIFT) | ((uint64_t)(f ? 1 : 0) << ITEM_CONT_SHIFT);
      }
    });
  }
  flush(g >> 5);
}

}  // namespace

extern "C" int pengk_pack(const uint8_t* codes, const int64_t* offs, int64_t n_seq, int W, int item_windows,
                          pengk_packed* out) {
  if (!out) return fail(PENGK_ERR_ARG, "pengk_pack: out is NULL");
  memset(out, 0, sizeof *out);
  if (n_seq < 0 || (n_seq > 0 && (!codes || !offs))) return fail(PENGK_ERR_ARG, "pengk_pack: NULL input");
  if (!valid_w(W)) return fail(PENGK_ERR_ARG, "pattern length %d unsupported (even, %d..%d)", W, PENGK_MIN_W, PENGK_MAX_W);
  if (item_windows == 0) item_windows = PENGK_DEFAULT_ITEM_WINDOWS;
  if (item_windows < PENGK_MIN_ITEM_WINDOWS || item_windows > 65535)
    return fail(PENGK_ERR_ARG, "item_windows %d out of range [%d,65535]", item_windows, PENGK_MIN_ITEM_WINDOWS);
  const uint64_t M = (uint64_t)item_windows;

  // contiguous sequence ranges of roughly equal size in bases
  const uint64_t total = n_seq ? (uint64_t)(offs[n_seq] - offs[0]) : 0;
  unsigned hw = std::thread::hardware_concurrency();
  if (hw == 0) hw = 1;
  if (hw > 32) hw = 32;
  unsigned nt = total < (1u << 22) ? 1u : hw;
  if (const char* e = getenv("PENGK_PACK_THREADS")) {  // tests: force a thread count
    const int v = atoi(e);
    if (v >= 1 && v <= 64) nt = (unsigned)v;
  }
  if ((int64_t)nt > n_seq) nt = n_seq > 0 ? (unsigned)n_seq : 1u;
  std::vector<int64_t> cut(nt + 1, n_seq);
  cut[0] = 0;
  {
    int64_t s = 0;
    for (unsigned t = 1; t < nt; ++t) {
      const int64_t target = offs[0] + (int64_t)(total * t / nt);
      while (s < n_seq && offs[s] < target) ++s;
      cut[t] = s;
    }
  }
  std::vector<RangeStat> st(nt);
  try {
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t)
      th.emplace_back(measure_range, codes, offs, cut[t], cut[t + 1], W, M, std::ref(st[t]));
    measure_range(codes, offs, cut[0], cut[1], W, M, st[0]);
    for (auto& x : th) x.join();
  } catch (const std::exception&) {
    return fail(PENGK_ERR_NOMEM, "pengk_pack: cannot start host threads");
  }

  uint64_t n_bases = 0, n_windows = 0, n_items = 0, bound = 0, max_len = 0;
  int all_whole = 1;
  std::vector<uint64_t> base0(nt), item0(nt);
  for (unsigned t = 0; t < nt; ++t) {
    base0[t] = PENGK_FRONT_PAD_BASES + n_bases;
    item0[t] = n_items;
    n_bases += st[t].bases;
    n_windows += st[t].windows;
    n_items += st[t].items;
    bound += st[t].bound;
    if (st[t].max_len > max_len) max_len = st[t].max_len;
    all_whole &= st[t].all_whole;
    for (int i = 0; i < 84; ++i) out->bg_counts[i] += st[t].bg[i];
  }
  if (PENGK_FRONT_PAD_BASES + n_bases > ITEM_WS_MASK) return fail(PENGK_ERR_RANGE, "packed stream exceeds 2^40 bases; shard the input");

  const uint64_t n_words = (PENGK_FRONT_PAD_BASES + n_bases + 31) / 32 + 4;  // >= 96 zero bases behind the data
  out->words = (uint64_t*)calloc(n_words, sizeof(uint64_t));
  out->items = (uint64_t*)malloc((n_items ? n_items : 1) * sizeof(uint64_t));
  if (!out->words || !out->items) {
    free(out->words);
    free(out->items);
    memset(out, 0, sizeof *out);
    return fail(PENGK_ERR_NOMEM, "pengk_pack: out of host memory");
  }
  {
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t)
      th.emplace_back(write_range, codes, offs, cut[t], cut[t + 1], W, M, base0[t], item0[t], out->words, out->items);
    write_range(codes, offs, cut[0], cut[1], W, M, base0[0], item0[0], out->words, out->items);
    for (auto& x : th) x.join();
  }
  out->n_words = n_words;
  out->n_items = n_items;
  out->n_bases = n_bases;
  out->n_windows = n_windows;
  out->max_bin_bound = bound;
  out->n_sequences = (uint64_t)n_seq;
  out->max_len = max_len;
  out->W = W;
  out->item_windows = item_windows;
  out->all_whole = all_whole;
  return PENGK_OK;
}

extern "C" void pengk_packed_free(pengk_packed* p) {
  if (!p) return;
  free(p->words);
  free(p->items);
  p->words = nullptr;
  p->items = nullptr;
  p->n_words = p->n_items = 0;
}
