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
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include "pengk_internal.h"

using namespace pengk;

namespace {

struct BitWriter {
  std::vector<uint64_t>& w;
  uint64_t nbases;  // bases written so far (incl. front pad)
  explicit BitWriter(std::vector<uint64_t>& words) : w(words), nbases(0) {}
  inline void put(unsigned d) {
    const uint64_t word = nbases >> 5;
    if (word >= w.size()) w.resize(w.size() ? w.size() * 2 : 1024, 0);
    w[word] |= (uint64_t)d << (2 * (nbases & 31));
    ++nbases;
  }
};

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

  std::vector<uint64_t> words;
  std::vector<uint64_t> items;
  try {
    const uint64_t total = n_seq ? (uint64_t)(offs[n_seq] - offs[0]) : 0;
    words.assign((PENGK_FRONT_PAD_BASES + total + 31) / 32 + 4, 0);
    items.reserve((size_t)n_seq + 16);
  } catch (const std::bad_alloc&) {
    return fail(PENGK_ERR_NOMEM, "pengk_pack: out of host memory");
  }
  BitWriter bw(words);
  bw.nbases = PENGK_FRONT_PAD_BASES;

  uint64_t n_windows = 0, bound = 0, max_len = 0;
  int all_whole = 1;
  int64_t* bg = out->bg_counts;
  int64_t* bgk[3] = {bg, bg + 4, bg + 20};

  try {
    for (int64_t s = 0; s < n_seq; ++s) {
      const uint8_t* seq = codes + offs[s];
      const int64_t L = offs[s + 1] - offs[s];
      if ((uint64_t)L > max_len) max_len = (uint64_t)L;

      // background (k+1)-mer counts with the reference's invalid-base behaviour
      {
        uint32_t d9 = 0, m9 = 0;
        for (int64_t i = 0; i < L; ++i) {
          const unsigned c = seq[i];
          const unsigned inv = (c == 0 || c > 4);
          d9 = ((d9 << 2) | (inv ? 0u : c - 1u)) & 0x3FFFFu;
          m9 = ((m9 << 1) | inv) & 0x1FFu;
          for (int k = 0; k <= 2; ++k) {
            if (i < k) break;
            const uint32_t y = d9 & ((1u << (2 * (k + 1))) - 1u);
            if (m9 == 0 || y == 0) ++bgk[k][y];
          }
        }
      }

      // visited runs
      int64_t i = 0;
      int runs = 0;
      bool whole = false;
      while (i < L) {
        int64_t j = i;
        while (j < L && seq[j] >= 1 && seq[j] <= 4) ++j;
        const int64_t len = j - i;
        if (len >= W) {
          ++runs;
          whole = (i == 0 && j == L);
          const uint64_t start = bw.nbases;
          for (int64_t t = i; t < j; ++t) bw.put(seq[t] - 1u);
          const uint64_t nwin = (uint64_t)(len - W + 1);
          n_windows += nwin;
          bound += (nwin + W - 1) / W;
          for (uint64_t f = 0; f < nwin; f += (uint64_t)item_windows) {
            const uint64_t nw = nwin - f < (uint64_t)item_windows ? nwin - f : (uint64_t)item_windows;
            const uint64_t ws = start + f;
            if (ws > ITEM_WS_MASK) return fail(PENGK_ERR_RANGE, "packed stream exceeds 2^40 bases; shard the input");
            items.push_back(ws | (nw << ITEM_NW_SHIFT) | ((uint64_t)(f ? 1 : 0) << ITEM_CONT_SHIFT));
          }
          i = j + 2;
        } else {
          i = j + 1;
        }
      }
      if (!(runs == 1 && whole)) all_whole = 0;
    }
  } catch (const std::bad_alloc&) {
    return fail(PENGK_ERR_NOMEM, "pengk_pack: out of host memory");
  }

  const uint64_t n_words = (bw.nbases + 31) / 32 + 4;  // >= 96 zero bases behind the data
  out->words = (uint64_t*)calloc(n_words, sizeof(uint64_t));
  out->items = (uint64_t*)malloc((items.size() ? items.size() : 1) * sizeof(uint64_t));
  if (!out->words || !out->items) {
    free(out->words);
    free(out->items);
    memset(out, 0, sizeof *out);
    return fail(PENGK_ERR_NOMEM, "pengk_pack: out of host memory");
  }
  memcpy(out->words, words.data(), (size_t)((bw.nbases + 31) / 32) * sizeof(uint64_t));
  if (!items.empty()) memcpy(out->items, items.data(), items.size() * sizeof(uint64_t));
  out->n_words = n_words;
  out->n_items = items.size();
  out->n_bases = bw.nbases - PENGK_FRONT_PAD_BASES;
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
