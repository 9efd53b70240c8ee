// BasePattern -- host mirror; the constructor is the hot path: pack -> count (K1) -> mirror -> sweep
// (K2+K3) on the device, then one copy of each table into the host arrays the public getters expose
// (replaces src/base_pattern.cpp:17-64 and everything it calls, :231-441).
#include "base_pattern.h"

#include <sys/mman.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <string>
#include <iomanip>
#include <iostream>

#include "ranked_prefix.h"
#include "utils.h"

using pengk_host::check;
using pengk_host::context;

BasePattern::BasePattern(const size_t pattern_length, Strand s, const int k, const int max_k, SequenceSet* sequence_set,
                         BackgroundModel* bg) {
  alphabet_size = Alphabet::getSize();
  strand = s;
  background_model = bg;
  init(pattern_length);
  number_patterns = factor[pattern_length];
  n_sequences = sequence_set->getN();
  this->k = k;
  this->max_k = std::max(k, max_k);
  const int W = (int)pattern_length;
  const int both = strand == Strand::BOTH_STRANDS;
  const size_t NP = number_patterns;

  // ---- sequences -> 2-bit stream + scan items (host packer resolves the N / skip scan rule) ----------
  pengk_host::Lap lap("    ");
  // multi-GPU run: the sequence set of this rank holds its shard of whole records only (sharded ingest,
  // shared/SequenceSet.h); the tables are summed below
  pengk_host::DeviceBuffer<uint64_t> d_words, d_items, d_ltot(1);
  if (const pengk_host::PackedInput* in = pengk_host::packed_input(sequence_set, W)) {
    // the CLI's input set: packed and uploaded chunk by chunk while the FASTA file was read (host/device.cpp)
    check(pengk_set_sequences(context(), in->d_words, in->n_words, in->d_items, in->n_items, W, in->item_windows, in->max_bin_bound,
                              in->all_whole),
          "pengk_set_sequences");
    check(pengk_set_option(context(), "n_windows_hint", (int64_t)in->n_windows), "pengk_set_option");
    lap("attach (packed and uploaded during the read)");
  } else {
    pengk_packed pk;
    check(pengk_pack(sequence_set->codes(), sequence_set->offsets(), (int64_t)sequence_set->getLocalN(), W, 0, &pk), "pengk_pack");
    lap("pack");
    d_words.resize(pk.n_words);
    d_items.resize(pk.n_items + 1);
    d_words.upload(pk.words, pk.n_words);
    if (pk.n_items) d_items.upload(pk.items, pk.n_items);
    check(pengk_set_sequences(context(), d_words.get(), pk.n_words, d_items.get(), pk.n_items, W, pk.item_windows,
                              pk.max_bin_bound, pk.all_whole),
          "pengk_set_sequences");
    check(pengk_set_option(context(), "n_windows_hint", (int64_t)pk.n_windows), "pengk_set_option");
    pengk_packed_free(&pk);
    lap("upload");
  }

  // ---- K1 count (+ twin copy), K2+K3 sweep ---------------------------------------------------------------
  d_counts.resize(NP);
  check(pengk_count(context(), both, d_counts.get(), d_ltot.get()), "pengk_count");
  if (pengk_host::launched()) {
    // C1, the one exchange step: shard counts add exactly (the non-overlap rule is per sequence, src/base_pattern.cpp:382).
    // The 32-bit bins must hold the GLOBAL counts: the bound is all-reduced and checked first.
    check(pengk_comm_check_bin_bound(context()), "pengk_comm_check_bin_bound");
    check(pengk_allreduce_tables(context(), W, d_counts.get(), d_ltot.get(), nullptr), "pengk_allreduce_tables");
  }
  if (both) check(pengk_mirror_counts(context(), W, d_counts.get()), "pengk_mirror_counts");

  float hV[84] = {0};
  for (int o = 0, at = 0; o <= 2; at += 1 << (2 * (o + 1)), ++o)
    if (o <= bg->getOrder())
      for (int y = 0; y < (1 << (2 * (o + 1))); ++y) hV[at + y] = bg->getV()[o][y];
  pengk_host::DeviceBuffer<float> d_V(84);
  d_logp.resize(NP);
  d_z.resize(NP);
  d_V.upload(hV, 84);
  d_bgprob.resize((size_t)(this->max_k + 1) * NP);
  d_expected.resize(NP);
  check(pengk_pattern_stats(context(), W, both, this->k, this->max_k, d_V.get(), d_ltot.get(), d_counts.get(), d_bgprob.get(),
                            d_expected.get(), d_logp.get(), d_z.get()),
        "pengk_pattern_stats");

  if (lap.on) check(pengk_synchronize(context()), "pengk_synchronize");
  lap("count + sweep");
  uint64_t lt = 0;
  d_ltot.download(&lt, 1);
  ltot = lt;

  // Test hook (tests/test_gpu_multirank.py): rank 0 leaves the GLOBAL tables as files, so that a multi-rank run can be
  // compared with the compiled reference's per-shard sums (tests/golden/shard_prefix_checksums.json), not just with
  // this program's own single-process run.
  if (const char* dir = std::getenv("PENGK_DUMP_TABLES")) {
    if (pengk_host::rank() == 0) {
      auto dump = [&](const char* name, const void* p, size_t bytes) {
        const std::string path = std::string(dir) + "/" + name;
        FILE* f = fopen(path.c_str(), "wb");
        if (!f || fwrite(p, 1, bytes, f) != bytes) {
          std::cerr << "Error: PENGK_DUMP_TABLES: cannot write " << path << std::endl;
          exit(1);
        }
        fclose(f);
      };
      dump("counts.u32", host_counts32(), NP * sizeof(uint32_t));
      dump("z.f32", host_zscore(), NP * sizeof(float));
      dump("expected.f32", host_expected(), NP * sizeof(float));
      dump("V.f32", hV, sizeof hV);
      const std::string meta = "ltot " + std::to_string(ltot) + "\nN " + std::to_string(n_sequences) + "\n";
      dump("meta.txt", meta.data(), meta.size());
    }
  }
}

BasePattern::~BasePattern() {
  delete[] pattern_counter;
  const size_t table = number_patterns * sizeof(float);
  release_mirror(counts32, table);
  release_mirror(pattern_bg_probabilities ? pattern_bg_probabilities[0] : nullptr, (size_t)(max_k + 1) * table);
  release_mirror(pattern_logp, table);
  release_mirror(pattern_zscore, table);
  release_mirror(expected_counts, table);
  delete[] pattern_bg_probabilities;
  delete[] factor;
}

// ---- host mirrors behind the raw-pointer getters ---------------------------------------------------------------
// Small tables (W <= 10: 4 MB) are page-locked, the copy then runs at link speed.  Page-locking costs time in
// proportion to the size, and at W >= 12 (64 MB a table) more than the faster copy returns: those mirrors are plain
// huge-page memory.  (PENGK_MIRROR_PINNED_MAX_MB moves the limit: developer knob.)
static size_t pinned_limit() {
  static const size_t v = [] {
    const char* e = std::getenv("PENGK_MIRROR_PINNED_MAX_MB");
    return (size_t)(e ? std::atol(e) : 16) << 20;
  }();
  return v;
}

void* BasePattern::mirror(const void* d_src, size_t bytes) {
  void* h = nullptr;
  if (bytes <= pinned_limit()) {
    check(pengk_host_alloc(context(), bytes, &h), "pengk_host_alloc");
  } else {
    const size_t huge = (size_t)2 << 20;
    if (posix_memalign(&h, huge, (bytes + huge - 1) / huge * huge) != 0) throw std::bad_alloc();
    madvise(h, (bytes + huge - 1) / huge * huge, MADV_HUGEPAGE);
  }
  check(pengk_memcpy_d2h(context(), h, d_src, bytes), "pengk_memcpy_d2h");
  return h;
}

void BasePattern::release_mirror(void* h, size_t bytes) {
  if (!h) return;
  if (bytes <= pinned_limit())
    check(pengk_host_free(context(), h), "pengk_host_free");
  else
    std::free(h);
}

float* BasePattern::fetch(const float* d_src, size_t n) { return (float*)mirror(d_src, n * sizeof(float)); }

// the counts as the device holds them (32-bit): what the pipeline itself reads -- a few hundred entries of 4^W
const uint32_t* BasePattern::host_counts32() {
  if (!counts32) counts32 = (uint32_t*)mirror(d_counts.get(), number_patterns * sizeof(uint32_t));
  return counts32;
}

// the reference's size_t table (src/base_pattern.h:129), for callers of the public getter: widened on first use (at
// W = 12 that is 134 MB written for a table nobody in the pipeline reads whole)
size_t* BasePattern::host_counts() {
  if (!pattern_counter) {
    const uint32_t* c32 = host_counts32();
    pattern_counter = new size_t[number_patterns];
    for (size_t i = 0; i < number_patterns; ++i) pattern_counter[i] = c32[i];
  }
  return pattern_counter;
}

float** BasePattern::host_bgprob() {
  if (!pattern_bg_probabilities) {
    float* all = fetch(d_bgprob.get(), (size_t)(max_k + 1) * number_patterns);
    pattern_bg_probabilities = new float*[max_k + 1];
    for (int o = 0; o <= max_k; ++o) pattern_bg_probabilities[o] = all + (size_t)o * number_patterns;
  }
  return pattern_bg_probabilities;
}

float* BasePattern::host_logp() {
  if (!pattern_logp) pattern_logp = fetch(d_logp.get(), number_patterns);
  return pattern_logp;
}

float* BasePattern::host_zscore() {
  if (!pattern_zscore) pattern_zscore = fetch(d_z.get(), number_patterns);
  return pattern_zscore;
}

float* BasePattern::host_expected() {
  if (!expected_counts) expected_counts = fetch(d_expected.get(), number_patterns);
  return expected_counts;
}

void BasePattern::init(size_t pattern_length) {
  this->pattern_length = pattern_length;
  factor = new size_t[pattern_length + 1];
  for (size_t i = 0; i <= pattern_length; ++i) factor[i] = (size_t)1 << (2 * i);
}

std::string BasePattern::toString(size_t pattern_id) {
  std::string out;
  for (size_t p = 0; p < pattern_length; ++p) out += Alphabet::getBase((uint8_t)(getNucleotideAtPos(pattern_id, p) + 1));
  return out;
}

// reverse the digit order and complement every digit (3 - d)
size_t BasePattern::getRevCompId(const size_t pattern_id) {
  size_t r = 0, x = pattern_id;
  for (size_t p = 0; p < pattern_length; ++p) {
    r = (r << 2) | (3 - (x & 3));
    x >>= 2;
  }
  return r;
}

size_t BasePattern::getFastRevCompId(const size_t pattern_id) { return getRevCompId(pattern_id); }

size_t BasePattern::baseId2IUPACId(const size_t base_pattern) {
  size_t id = 0;
  for (size_t p = 0; p < pattern_length; ++p) id += getNucleotideAtPos(base_pattern, p) * IUPACPattern::iupac_factor[p];
  return id;
}

float BasePattern::getExpCountFraction(const size_t pattern, const size_t pseudo_expected_pattern_counts) {
  return (host_expected()[pattern] + (float)pseudo_expected_pattern_counts) / (float)(size_t)host_counts32()[pattern];
}

float BasePattern::getMutualInformationScore(const size_t pattern) {
  const unsigned int observed = (unsigned int)host_counts32()[pattern];
  return mutual_information_score((float)observed, host_expected()[pattern], (unsigned int)n_sequences);
}

float BasePattern::getOptimizationScore(const OPTIMIZATION_SCORE score_type, const size_t pattern,
                                        const size_t pseudo_expected_pattern_counts) {
  switch (score_type) {
    case OPTIMIZATION_SCORE::kLogPval: return getLogPval(pattern);
    case OPTIMIZATION_SCORE::kExpCounts: return getExpCountFraction(pattern, pseudo_expected_pattern_counts);
    case OPTIMIZATION_SCORE::MutualInformation: return getMutualInformationScore(pattern);
  }
  std::cerr << "Error: unknown score type!" << std::endl;
  exit(1);
}

// Seeds in descending z order.  The same non-stable std::sort over the same bit-identical z array as the
// reference (src/base_pattern.cpp:458) decides which reverse-complement twin of a tie comes first.
std::vector<size_t> BasePattern::select_base_patterns(const float zscore_threshold, const size_t count_threshold,
                                                      bool single_stranded, bool filter_neighbors) {
  std::vector<size_t> selected;
  std::vector<char> seen(number_patterns, 0);
  ranked_prefix::EntryVec order;
  size_t n_ranked = 0;
  static const bool on_device = [] {
    const char* e = std::getenv("PENGK_SEED_SELECT");
    return e && std::string(e) == "device";
  }();
  if (on_device) {
    // Opt-in fast mode: the candidates (z >= threshold, count >= threshold) are compacted on the
    // device and ranked here by (z descending, id ascending) -- no 4^W tables cross the link, no 4^W-entry ranking.
    // DOCUMENTED DIFFERENCE: exact z ties (every reverse-complement pair under both strands) are ordered by id, where
    // the reference leaves them in the order of its non-stable std::sort; the seed set is the same up to the strand
    // a pair is named on (tests/test_gpu_cli.py).
    std::vector<uint32_t> ids(1 << 16);
    std::vector<float> zs(ids.size());
    int64_t n = 0;
    for (;;) {
      check(pengk_seed_candidates(context(), (int)pattern_length, d_z.get(), d_counts.get(), zscore_threshold, count_threshold,
                                  ids.data(), zs.data(), (int64_t)ids.size(), &n),
            "pengk_seed_candidates");
      if ((size_t)n <= ids.size()) break;
      ids.resize((size_t)n);
      zs.resize((size_t)n);
    }
    order.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) order[i] = ranked_prefix::Entry(zs[i], ids[i]);
    std::sort(order.begin(), order.end(), [](const ranked_prefix::Entry& a, const ranked_prefix::Entry& b) {
      return a.z > b.z || (a.z == b.z && a.id < b.id);
    });
    n_ranked = order.size();
  }
  // default: the ranking std::sort(order, sort_indices(pattern_zscore)) would leave (:458), down to the threshold only
  pengk_host::Lap lap("    ");
  const float* pattern_zscore = on_device ? nullptr : host_zscore();
  const uint32_t* pattern_counter = on_device ? nullptr : host_counts32();
  lap("z-scores and counts to the host");
  if (!on_device) n_ranked = ranked_prefix::rank(pattern_zscore, number_patterns, zscore_threshold, order);
  lap("ranking down to the threshold");
  for (size_t r = 0; r < n_ranked; ++r) {
    const size_t x = order[r].id;
    if (!on_device) {
      if (pattern_zscore[x] < zscore_threshold) break;
      if (pattern_counter[x] < count_threshold) continue;
    }
    if (seen[x] || (!single_stranded && seen[getFastRevCompId(x)])) continue;
    selected.push_back(x);
    seen[x] = 1;
    if (filter_neighbors)
      for (size_t p = 0; p < pattern_length; ++p) {
        const size_t masked = x & ~((size_t)3 << (2 * p));
        for (size_t c = 0; c < 4; ++c) seen[masked | (c << (2 * p))] = 1;
      }
  }
  lap("walk");
  return selected;
}

std::vector<size_t> BasePattern::generate_double_stranded_em_optimization_patterns() {
  std::vector<size_t> out;
  for (size_t x = 0; x < number_patterns; ++x)
    if (x <= getFastRevCompId(x)) out.push_back(x);
  return out;
}

void BasePattern::print_patterns(std::vector<size_t> patterns) {
  std::cout << std::setw(15) << "pattern" << "\t" << std::setw(15) << "observed" << "\t" << std::setw(15) << "enrichment"
            << "\t" << std::setw(15) << "zscore" << std::endl
            << std::endl;
  std::cout << std::fixed << std::setprecision(2);
  const uint32_t* counts = host_counts32();
  const float* expected_counts = host_expected();
  const float* pattern_zscore = host_zscore();
  for (size_t x : patterns) {
    const size_t c = counts[x];  // (size_t like the reference's table: c / expected converts it to float the same way)
    std::cout << std::setw(15) << toString(x) << "\t" << std::setw(15) << c << "\t" << std::setw(15) << (c / expected_counts[x]) << "\t"
              << std::setw(15) << pattern_zscore[x] << std::endl;
  }
}
