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
#include <sys/mman.h>
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

// ---- fast path: every sequence is one whole visited run (no invalid base, L >= W) ----------------------------
// Then the stream is the 2-bit image of the contiguous code bytes and a sequence's stream position is known
// without a prefix sum, so ONE pass does validity check, packing and the background counters, 8 bases per step.
// Any other sequence makes the range report failure and pengk_pack restarts on the general two-pass path.
inline uint64_t load8(const uint8_t* p) {
  uint64_t v;
  memcpy(&v, p, 8);
  return v;
}
// 8 code bytes (1..4, first base in the low byte) -> nonzero iff one of them is not a valid code
inline uint64_t invalid8(uint64_t v) {
  const uint64_t zero = (v - 0x0101010101010101ull) & ~v & 0x8080808080808080ull;          // a byte == 0
  const uint64_t big = ((v + 0x7B7B7B7B7B7B7B7Bull) | v) & 0x8080808080808080ull;          // a byte >= 5
  return zero | big;  // exact when all bytes are valid; any invalid byte gives nonzero (carries only add flags)
}
// 8 valid code bytes -> 16 bits, base j in bits [2j, 2j+2)
inline uint32_t pack8(uint64_t v) {
  uint64_t t = v - 0x0101010101010101ull;                 // digits 0..3 per byte
  t = (t | (t >> 6)) & 0x000F000F000F000Full;             // 2 digits per 16-bit lane
  t = (t | (t >> 12)) & 0x000000FF000000FFull;            // 4 digits per 32-bit lane
  return (uint32_t)((t | (t >> 24)) & 0xFFFFu);
}

struct FastStat {
  uint64_t windows = 0, items = 0, bound = 0, max_len = 0;
  uint64_t h3[64] = {0};  // 3-mers ending at i >= 2, index = first base in the LOW digit
  uint64_t e1[16] = {0};  // the 2-mer ending at i == 1 of every sequence (BaMM index: first base high)
  uint64_t e0[4] = {0};   // the base at i == 0
  int ok = 1;
};

void fast_range(const uint8_t* codes, const int64_t* offs, int64_t s0, int64_t s1, int W, uint64_t M, uint64_t g0, uint64_t* words,
                FastStat& st) {
  uint64_t g = g0;    // stream position of the next base
  uint64_t acc = 0;   // bits gathered for word g >> 5
  bool shared = true; // the first and the last word of the range may be shared with a neighbour range
  auto put = [&](uint64_t word) {
    if (shared) {
      if (acc) __atomic_fetch_or(&words[word], acc, __ATOMIC_RELAXED);
      shared = false;
    } else {
      words[word] = acc;
    }
    acc = 0;
  };
  for (int64_t s = s0; s < s1; ++s) {
    const uint8_t* seq = codes + offs[s];
    const int64_t L = offs[s + 1] - offs[s];
    if (L < W) {
      st.ok = 0;
      return;
    }
    if ((uint64_t)L > st.max_len) st.max_len = (uint64_t)L;
    uint64_t bad = 0;
    uint32_t carry = 0;  // the last two bases seen (4 bits)
    int64_t i = 0;
    for (; i + 8 <= L; i += 8) {
      const uint64_t v = load8(seq + i);
      bad |= invalid8(v);
      const uint32_t t = pack8(v);
      const unsigned sh = 2u * (unsigned)(g & 31);
      acc |= (uint64_t)t << sh;
      g += 8;
      if (sh + 16 >= 64) {
        put((g >> 5) - 1);
        if (sh + 16 > 64) acc = (uint64_t)t >> (64 - sh);
      }
      const uint32_t x = carry | (t << 4);  // base j of the chunk sits at bits [2j+4, 2j+6)
      if (i) {
        ++st.h3[x & 63];
        ++st.h3[(x >> 2) & 63];
      }
      ++st.h3[(x >> 4) & 63];
      ++st.h3[(x >> 6) & 63];
      ++st.h3[(x >> 8) & 63];
      ++st.h3[(x >> 10) & 63];
      ++st.h3[(x >> 12) & 63];
      ++st.h3[(x >> 14) & 63];
      carry = t >> 12;
    }
    for (; i < L; ++i) {
      const unsigned c = seq[i];
      bad |= (c == 0 || c > 4);
      const uint32_t d = (c - 1u) & 3u;
      acc |= (uint64_t)d << (2 * (g & 31));
      ++g;
      if ((g & 31) == 0) put((g >> 5) - 1);
      if (i >= 2) ++st.h3[carry | (d << 4)];
      carry = (carry >> 2) | (d << 2);
    }
    if (bad) {
      st.ok = 0;
      return;
    }
    // the two incomplete k-mers at the head of the sequence (L >= W >= 4)
    const unsigned b0 = seq[0] - 1u, b1 = seq[1] - 1u;
    ++st.e0[b0];
    ++st.e1[b0 * 4 + b1];
    const uint64_t nwin = (uint64_t)(L - W + 1);
    st.windows += nwin;
    st.bound += (nwin + W - 1) / W;
    st.items += (nwin + M - 1) / M;
  }
  shared = true;
  put(g >> 5);
}

void fast_items(const int64_t* offs, int64_t s0, int64_t s1, int W, uint64_t M, uint64_t item0, uint64_t* items) {
  uint64_t it = item0;
  for (int64_t s = s0; s < s1; ++s) {
    const uint64_t run0 = PENGK_FRONT_PAD_BASES + (uint64_t)(offs[s] - offs[0]);
    const uint64_t nwin = (uint64_t)(offs[s + 1] - offs[s] - W + 1);
    for (uint64_t f = 0; f < nwin; f += M) {
      const uint64_t nw = nwin - f < M ? nwin - f : M;
      items[it++] = (run0 + f) | (nw << ITEM_NW_SHIFT) | ((uint64_t)(f ? 1 : 0) << ITEM_CONT_SHIFT);
    }
  }
}

// Threads that are always joined: if starting one throws (std::system_error) while others run, unwinding would
// destroy joinable std::thread objects -- std::terminate before any catch.  The destructor joins what was started.
struct ThreadGroup {
  std::vector<std::thread> th;
  template <class... A>
  void spawn(A&&... a) { th.emplace_back(std::forward<A>(a)...); }
  void join() {
    for (auto& x : th)
      if (x.joinable()) x.join();
  }
  ~ThreadGroup() { join(); }
};

template <class F>
void run_ranges(unsigned nt, F&& f) {
  ThreadGroup g;
  for (unsigned t = 1; t < nt; ++t) g.spawn(f, t);
  f(0u);
  g.join();
}

// The two output arrays (0.5 GB + 80 MB for 10M sequences) are written exactly once, by all threads: zero-filled
// anonymous memory on transparent huge pages, so that the first touch costs one fault per 2 MiB instead of one per 4 KiB
// (~10 us each inside a VM: the faults were half of the packer's time).  A 32-byte header in front of the block says how
// to release it.  PENGK_NO_HUGEPAGES=1 keeps ordinary pages.
struct BlockHeader {
  void* base;
  size_t mapped;  // 0: base came from calloc
  uint64_t pad[2];
};
void* block_alloc_zero(size_t bytes) {
  const size_t huge = (size_t)2 << 20;
  if (bytes >= 2 * huge) {
    const size_t mapped = (bytes + sizeof(BlockHeader) + 2 * huge - 1) / huge * huge;
    void* base = mmap(nullptr, mapped, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (base != MAP_FAILED) {
      static const bool want = getenv("PENGK_NO_HUGEPAGES") == nullptr;
      if (want) madvise(base, mapped, MADV_HUGEPAGE);
      // the data starts on a 2 MiB boundary, the header sits right in front of it
      char* data = (char*)(((uintptr_t)base + sizeof(BlockHeader) + huge - 1) / huge * huge);
      BlockHeader* h = (BlockHeader*)(data - sizeof(BlockHeader));
      h->base = base;
      h->mapped = mapped;
      return data;
    }
  }
  char* base = (char*)calloc(1, bytes + sizeof(BlockHeader));
  if (!base) return nullptr;
  BlockHeader* h = (BlockHeader*)base;
  h->base = base;
  h->mapped = 0;
  return base + sizeof(BlockHeader);
}
void block_free(void* p) {
  if (!p) return;
  BlockHeader* h = (BlockHeader*)((char*)p - sizeof(BlockHeader));
  if (h->mapped) munmap(h->base, h->mapped);
  else free(h->base);
}

struct FreeGuard {  // scratch released on every exit path unless handed over
  void* p;
  ~FreeGuard() { block_free(p); }
};

// the whole fast path; returns 0 when the input needs the general path (out untouched except for scratch it frees)
int pack_fast(const uint8_t* codes, const int64_t* offs, int64_t n_seq, int W, int item_windows, unsigned nt,
              const std::vector<int64_t>& cut, uint64_t total, pengk_packed* out) {
  const uint64_t M = (uint64_t)item_windows;
  if (PENGK_FRONT_PAD_BASES + total > ITEM_WS_MASK) return 0;  // the general path reports the range error
  const uint64_t n_words = (PENGK_FRONT_PAD_BASES + total + 31) / 32 + 4;
  uint64_t* words = (uint64_t*)block_alloc_zero(n_words * sizeof(uint64_t));
  if (!words) return 0;
  FreeGuard words_guard{words};  // a failing thread start below must not leak the stream
  std::vector<FastStat> st(nt);
  run_ranges(nt, [&](unsigned t) {
    fast_range(codes, offs, cut[t], cut[t + 1], W, M, PENGK_FRONT_PAD_BASES + (uint64_t)(offs[cut[t]] - offs[0]), words, st[t]);
  });
  uint64_t n_windows = 0, n_items = 0, bound = 0, max_len = 0;
  std::vector<uint64_t> item0(nt);
  uint64_t h3[64] = {0}, e1[16] = {0}, e0[4] = {0};
  for (unsigned t = 0; t < nt; ++t) {
    if (!st[t].ok) return 0;  // (the guard frees the stream)
    item0[t] = n_items;
    n_windows += st[t].windows;
    n_items += st[t].items;
    bound += st[t].bound;
    if (st[t].max_len > max_len) max_len = st[t].max_len;
    for (int i = 0; i < 64; ++i) h3[i] += st[t].h3[i];
    for (int i = 0; i < 16; ++i) e1[i] += st[t].e1[i];
    for (int i = 0; i < 4; ++i) e0[i] += st[t].e0[i];
  }
  uint64_t* items = (uint64_t*)block_alloc_zero((n_items ? n_items : 1) * sizeof(uint64_t));
  if (!items) return 0;
  FreeGuard items_guard{items};
  run_ranges(nt, [&](unsigned t) { fast_items(offs, cut[t], cut[t + 1], W, M, item0[t], items); });

  // background counters (BaMM index: first base in the HIGH digit) from the 3-mer histogram and the head k-mers
  int64_t* bg0 = out->bg_counts;
  int64_t* bg1 = out->bg_counts + 4;
  int64_t* bg2 = out->bg_counts + 20;
  for (int x = 0; x < 64; ++x) {
    const int a = x & 3, b = (x >> 2) & 3, c = x >> 4;  // a = first base
    bg2[a * 16 + b * 4 + c] += (int64_t)h3[x];
    bg1[b * 4 + c] += (int64_t)h3[x];
  }
  for (int y = 0; y < 16; ++y) bg1[y] += (int64_t)e1[y];
  for (int y = 0; y < 16; ++y) bg0[y & 3] += bg1[y];
  for (int b = 0; b < 4; ++b) bg0[b] += (int64_t)e0[b];

  out->words = words;
  out->items = items;
  words_guard.p = nullptr;  // handed over
  items_guard.p = nullptr;
  out->n_words = n_words;
  out->n_items = n_items;
  out->n_bases = total;
  out->n_windows = n_windows;
  out->max_bin_bound = bound;
  out->n_sequences = (uint64_t)n_seq;
  out->max_len = max_len;
  out->W = W;
  out->item_windows = item_windows;
  out->all_whole = 1;
  return 1;
}

}  // namespace

extern "C" int pengk_pack(const uint8_t* codes, const int64_t* offs, int64_t n_seq, int W, int item_windows,
                          pengk_packed* out) {
  return pengk_pack_threads(codes, offs, n_seq, W, item_windows, 0, out);
}

extern "C" int pengk_pack_threads(const uint8_t* codes, const int64_t* offs, int64_t n_seq, int W, int item_windows,
                                  int threads, pengk_packed* out) {
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
  if (threads < 0 || threads > 64) return fail(PENGK_ERR_ARG, "pengk_pack_threads: %d threads (0 = automatic, 1..64)", threads);
  if (threads > 0) {
    nt = (unsigned)threads;
  } else if (const char* e = getenv("PENGK_PACK_THREADS")) {  // tests: force a thread count
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
  try {
    const char* g = getenv("PENGK_PACK_GENERAL");  // tests: compare the two paths
    if (n_seq > 0 && !(g && atoi(g)) && pack_fast(codes, offs, n_seq, W, item_windows, nt, cut, total, out)) return PENGK_OK;
  } catch (const std::exception&) {
    return fail(PENGK_ERR_NOMEM, "pengk_pack: cannot start host threads");
  }
  memset(out, 0, sizeof *out);
  std::vector<RangeStat> st(nt);
  try {
    ThreadGroup g;
    for (unsigned t = 1; t < nt; ++t) g.spawn(measure_range, codes, offs, cut[t], cut[t + 1], W, M, std::ref(st[t]));
    measure_range(codes, offs, cut[0], cut[1], W, M, st[0]);
    g.join();
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
  out->words = (uint64_t*)block_alloc_zero(n_words * sizeof(uint64_t));
  out->items = (uint64_t*)block_alloc_zero((n_items ? n_items : 1) * sizeof(uint64_t));
  if (!out->words || !out->items) {
    block_free(out->words);
    block_free(out->items);
    memset(out, 0, sizeof *out);
    return fail(PENGK_ERR_NOMEM, "pengk_pack: out of host memory");
  }
  try {
    ThreadGroup g;
    for (unsigned t = 1; t < nt; ++t)
      g.spawn(write_range, codes, offs, cut[t], cut[t + 1], W, M, base0[t], item0[t], out->words, out->items);
    write_range(codes, offs, cut[0], cut[1], W, M, base0[0], item0[0], out->words, out->items);
    g.join();
  } catch (const std::exception&) {
    block_free(out->words);
    block_free(out->items);
    memset(out, 0, sizeof *out);
    return fail(PENGK_ERR_NOMEM, "pengk_pack: cannot start host threads");
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

// ---- packing into buffers that collect the chunks of one input (the CLI's streaming ingest) -----------------------
// One thread per call, any number of calls at once: a chunk reserves its place with two atomic cursors -- words by the
// upper bound its bases give (exact when every sequence is one whole run), items by their exact number, known before
// the first item is written on both paths -- and writes absolute stream offsets, so the collection is attached to the
// device as it stands.  No allocation per chunk: the buffers are the caller's, zero-filled once.
extern "C" int pengk_pack_append(const uint8_t* codes, const int64_t* offs, int64_t n_seq, int W, int item_windows,
                                 pengk_pack_target* tg, pengk_packed* out) {
  if (!out || !tg || !tg->words || !tg->items) return fail(PENGK_ERR_ARG, "pengk_pack_append: NULL argument");
  memset(out, 0, sizeof *out);
  if (n_seq < 0 || (n_seq > 0 && (!codes || !offs))) return fail(PENGK_ERR_ARG, "pengk_pack_append: NULL input");
  if (!valid_w(W)) return fail(PENGK_ERR_ARG, "pattern length %d unsupported (even, %d..%d)", W, PENGK_MIN_W, PENGK_MAX_W);
  if (item_windows == 0) item_windows = PENGK_DEFAULT_ITEM_WINDOWS;
  if (item_windows < PENGK_MIN_ITEM_WINDOWS || item_windows > 65535)
    return fail(PENGK_ERR_ARG, "item_windows %d out of range [%d,65535]", item_windows, PENGK_MIN_ITEM_WINDOWS);
  const uint64_t M = (uint64_t)item_windows;
  const uint64_t total = n_seq ? (uint64_t)(offs[n_seq] - offs[0]) : 0;
  const uint64_t n_words = (PENGK_FRONT_PAD_BASES + total + 31) / 32 + 4;  // >= 96 zero bases behind the data
  const uint64_t w0 = __atomic_fetch_add(&tg->word_cursor, n_words, __ATOMIC_RELAXED);
  if (w0 + n_words > tg->words_cap) return fail(PENGK_ERR_RANGE, "pengk_pack_append: the word buffer is full");
  if ((w0 + n_words) * 32 > ITEM_WS_MASK) return fail(PENGK_ERR_RANGE, "packed stream exceeds 2^40 bases; shard the input");
  const uint64_t g0 = w0 * 32 + PENGK_FRONT_PAD_BASES;  // stream position of the chunk's first base
  out->n_words = n_words;
  out->words = tg->words + w0;  // (borrowed: not to be released)
  out->n_sequences = (uint64_t)n_seq;
  out->W = W;
  out->item_windows = item_windows;

  const char* force_general = getenv("PENGK_PACK_GENERAL");  // tests: compare the two paths
  if (n_seq > 0 && !(force_general && atoi(force_general))) {
    FastStat st;
    fast_range(codes, offs, 0, n_seq, W, M, g0, tg->words, st);
    if (st.ok) {
      const uint64_t i0 = __atomic_fetch_add(&tg->item_cursor, st.items, __ATOMIC_RELAXED);
      if (i0 + st.items > tg->items_cap) return fail(PENGK_ERR_RANGE, "pengk_pack_append: the item buffer is full");
      uint64_t it = i0;
      for (int64_t s = 0; s < n_seq; ++s) {
        const uint64_t run0 = g0 + (uint64_t)(offs[s] - offs[0]);
        const uint64_t nwin = (uint64_t)(offs[s + 1] - offs[s] - W + 1);
        for (uint64_t f = 0; f < nwin; f += M) {
          const uint64_t nw = nwin - f < M ? nwin - f : M;
          tg->items[it++] = (run0 + f) | (nw << ITEM_NW_SHIFT) | ((uint64_t)(f ? 1 : 0) << ITEM_CONT_SHIFT);
        }
      }
      int64_t* bg0 = out->bg_counts;
      int64_t* bg1 = out->bg_counts + 4;
      int64_t* bg2 = out->bg_counts + 20;
      for (int x = 0; x < 64; ++x) {
        const int a = x & 3, b = (x >> 2) & 3, c = x >> 4;  // a = first base
        bg2[a * 16 + b * 4 + c] += (int64_t)st.h3[x];
        bg1[b * 4 + c] += (int64_t)st.h3[x];
      }
      for (int y = 0; y < 16; ++y) bg1[y] += (int64_t)st.e1[y];
      for (int y = 0; y < 16; ++y) bg0[y & 3] += bg1[y];
      for (int b = 0; b < 4; ++b) bg0[b] += (int64_t)st.e0[b];
      out->items = tg->items + i0;
      out->n_items = st.items;
      out->n_bases = total;
      out->n_windows = st.windows;
      out->max_bin_bound = st.bound;
      out->max_len = st.max_len;
      out->all_whole = 1;
      return PENGK_OK;
    }
    memset(tg->words + w0, 0, n_words * sizeof(uint64_t));  // what the abandoned fast pass wrote
  }
  RangeStat st;
  measure_range(codes, offs, 0, n_seq, W, M, st);
  const uint64_t i0 = __atomic_fetch_add(&tg->item_cursor, st.items, __ATOMIC_RELAXED);
  if (i0 + st.items > tg->items_cap) return fail(PENGK_ERR_RANGE, "pengk_pack_append: the item buffer is full");
  write_range(codes, offs, 0, n_seq, W, M, g0, i0, tg->words, tg->items);
  for (int i = 0; i < 84; ++i) out->bg_counts[i] = st.bg[i];
  out->items = tg->items + i0;
  out->n_items = st.items;
  out->n_bases = st.bases;
  out->n_windows = st.windows;
  out->max_bin_bound = st.bound;
  out->max_len = st.max_len;
  out->all_whole = st.all_whole;
  return PENGK_OK;
}

extern "C" void pengk_packed_free(pengk_packed* p) {
  if (!p) return;
  block_free(p->words);
  block_free(p->items);
  p->words = nullptr;
  p->items = nullptr;
  p->n_words = p->n_items = 0;
}
