// Peng -- host mirror of the reference's orchestration (src/peng.cpp); see peng.h.
#include "peng.h"

#include <algorithm>
#include <cstdlib>
#include <climits>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <limits>
#include <map>

#include "base_pattern.h"
#include "device.h"
#include "iupac_alphabet.h"
#include "utils.h"

namespace {
void print_status(const std::string& header, bool leading_newline = true) {
  if (leading_newline) std::cout << std::endl;
  std::cout << "[STATUS] " << header << ":" << std::endl;
}

// add epsilon = 10^-precision / (1 - 4 * 10^-precision) to every cell and renormalise; mutates the PWM,
// so writing MEME and JSON in one run applies it twice (reference behaviour)
void no_zero_pwm(float** pwm, unsigned rows, unsigned precision) {
  const float delta = (float)std::pow(10, -static_cast<int>(precision));
  const float epsilon = delta / (1 - 4 * delta);
  for (unsigned i = 0; i < rows; ++i)
    for (unsigned j = 0; j < 4; ++j) pwm[i][j] += epsilon;
  IUPACPattern::normalize_pwm((int)rows, pwm);
}

void print_row(IUPACPattern* p, size_t W, float score, bool with_score) {
  std::cout << "\t" << std::setw(15) << IUPACPattern::toString(p->get_pattern(), W) << "\t" << std::setw(10) << p->get_sites()
            << "\t" << std::setw(5) << std::setprecision(2) << p->get_sites() / p->getExpectedCounts();
  if (with_score) std::cout << "\t" << std::setw(10) << std::setprecision(6) << score;
  std::cout << std::endl;
}
}  // namespace

Peng::Peng(Strand s, const int k, const int max_opt_k, SequenceSet* sequence_set, BackgroundModel* bg) {
  const int max_iupac_pattern_length = (int)(std::log((double)SIZE_MAX) / std::log((double)IUPAC_ALPHABET_SIZE) - 1);
  alphabet_size = Alphabet::getSize();
  this->k = k;
  max_k = std::max(k, max_opt_k);
  strand = s;
  n_sequences = sequence_set->getN();
  IUPACAlphabet::init(Alphabet::getAlphabet());
  IUPACPattern::init(max_iupac_pattern_length, bg->getV()[0]);
  bg_model = bg;
  this->sequence_set = sequence_set;
}

Peng::~Peng() {}

// ---- EM: all PWMs in one device call (replaces src/peng.cpp:48-197) --------------------------------------
void Peng::em_optimize_pwms(std::vector<IUPACPattern*>& patterns, BasePattern* base_patterns, float saturation_factor,
                            float min_em_threshold, int max_iterations, int background_order,
                            std::vector<IUPACPattern*>& optimized) {
  const size_t W = base_patterns->getPatternLength();
  const size_t n = patterns.size();
  std::vector<float> pw(n * W * 4);
  for (size_t i = 0; i < n; ++i)
    for (size_t p = 0; p < W; ++p)
      for (int a = 0; a < 4; ++a) pw[(i * W + p) * 4 + a] = patterns[i]->get_pwm()[p][a];
  const int n_ranks = pengk_host::world(), me = pengk_host::rank();
  if (n && !pengk_host::launched()) {
    pengk_host::check(pengk_em(pengk_host::context(), (int)W, (int64_t)n, pw.data(), saturation_factor, min_em_threshold,
                               max_iterations, base_patterns->device_counts(), base_patterns->device_bgprob(background_order),
                               nullptr, nullptr),
                      "pengk_em");
  } else if (n) {
    // multi-GPU: PWM i belongs to rank i mod n_ranks (PWMs are independent); an all-gather of equal blocks returns
    // every rank's results to every rank -- block r holds PWMs r, r + n_ranks, ... in that order, zero padded
    const size_t cell = W * 4, per = (n + (size_t)n_ranks - 1) / (size_t)n_ranks;
    std::vector<float> mine(per * cell, 0.0f), all((size_t)n_ranks * per * cell);
    size_t n_mine = 0;
    for (size_t i = (size_t)me; i < n; i += (size_t)n_ranks, ++n_mine)
      std::copy(pw.begin() + i * cell, pw.begin() + (i + 1) * cell, mine.begin() + n_mine * cell);
    if (n_mine)
      pengk_host::check(pengk_em(pengk_host::context(), (int)W, (int64_t)n_mine, mine.data(), saturation_factor, min_em_threshold,
                                 max_iterations, base_patterns->device_counts(),
                                 base_patterns->device_bgprob(background_order), nullptr, nullptr),
                        "pengk_em");
    pengk_host::DeviceBuffer<float> d_mine(per * cell), d_all((size_t)n_ranks * per * cell);
    d_mine.upload(mine.data(), per * cell);
    pengk_host::check(pengk_allgather(pengk_host::context(), d_mine.get(), d_all.get(), per * cell * sizeof(float)), "pengk_allgather");
    d_all.download(all.data(), all.size());
    for (size_t i = 0; i < n; ++i) {
      const size_t r = i % (size_t)n_ranks, k = i / (size_t)n_ranks;
      std::copy(all.begin() + (r * per + k) * cell, all.begin() + (r * per + k + 1) * cell, pw.begin() + i * cell);
    }
  }
  for (size_t i = 0; i < n; ++i) {
    std::vector<float*> rows(W);
    for (size_t p = 0; p < W; ++p) rows[p] = &pw[(i * W + p) * 4];
    IUPACPattern* opt = new IUPACPattern(patterns[i], rows.data());  // renormalises, builds the complement PWM
    optimized.push_back(opt);
    const float avg_info = calculate_pwm_info(opt->get_pwm(), (unsigned)W, (unsigned)alphabet_size) / W;
    std::cout << "em: " << IUPACPattern::toString(patterns[i]->get_pattern(), W) << " -> " << opt->get_pattern_string()
              << "   [ avg. info: " << std::setprecision(2) << avg_info << " ]" << std::endl;
  }
}

// ---- redundancy filter on the final list (src/peng.cpp:199-235) --------------------------------------------
void Peng::filter_redundancy(const float merge_bit_factor_threshold, std::vector<IUPACPattern*>& pats) {
  std::sort(pats.begin(), pats.end(), sort_IUPAC_patterns);
  std::vector<char> drop(pats.size(), 0);
  for (size_t i = 0; i < pats.size(); ++i) {
    if (drop[i]) continue;
    for (size_t j = i + 1; j < pats.size(); ++j) {
      if (drop[j] || pats[i]->get_pattern_length() != pats[j]->get_pattern_length()) continue;
      const int L = (int)pats[i]->get_pattern_length();
      const float s1 = IUPACPattern::calculate_s(pats[i]->get_pwm(), pats[j]->get_pwm(), bg_model->getV()[0], 0, 0, L);
      const float s2 = IUPACPattern::calculate_s(pats[i]->get_comp_pwm(), pats[j]->get_pwm(), bg_model->getV()[0], 0, 0, L);
      const float threshold = merge_bit_factor_threshold * L;
      if (s1 > threshold || s2 > threshold) {
        drop[j] = 1;
        break;  // the reference stops scanning partners of i after the first hit
      }
    }
  }
  std::vector<IUPACPattern*> kept;
  for (size_t i = 0; i < pats.size(); ++i) {
    if (drop[i]) delete pats[i];
    else kept.push_back(pats[i]);
  }
  pats.swap(kept);
}

// ---- greedy pairwise merging (src/peng.cpp:237-313) ----------------------------------------------------------
// The reference recomputes the similarity of EVERY pair after each merge (O(merges * n^2 * W * shifts) Jensen-Shannon
// terms, three log2 each).  Two things are kept instead:
//  * exact scores (the reference's arithmetic, IUPACPattern::calculate_S) per pair of motif serial numbers -- a pair's
//    score depends on the two motifs only;
//  * for sets of MERGE_GRID_MIN motifs and more, the device's similarity grid (pengk_motif_similarity: the same
//    scores in fp64, within ~1e-4 of the reference's float-rounded sums): a round needs the MAXIMUM over all pairs, so
//    only pairs whose device score lies within MERGE_MARGIN of the device maximum can be it, and only those are
//    evaluated exactly.  After a merge the new motif's column is the only part of the grid that is computed.
// Scan order and the strict `>` tie-break over the exactly evaluated pairs are the reference's, and every pair that
// could win is among them: decisions, printed motifs and scores are unchanged.
namespace {
constexpr size_t MERGE_GRID_MIN = 96;
// Margin around the device maximum within which a pair is evaluated exactly.  For motifs of up to 14 columns
// (the default max_merged_length): >> 3 sums x 56 terms x half a float ulp at |s| <= 32 (2e-4).  The reference's
// float-rounded sums have 4 L terms of magnitude up to ~2 L, so the bound grows with L^2: the margin is scaled by
// (L / 14)^2 for longer motifs (--max_merged_length up to PENGK_MAX_MOTIF_LEN = 64 columns: 0.042).
constexpr float MERGE_MARGIN = 2e-3f;
inline float merge_margin(size_t longest) {
  const float r = longest > 14 ? (float)longest / 14.0f : 1.0f;
  return MERGE_MARGIN * r * r;
}
}  // namespace

void Peng::merge_iupac_patterns(const size_t pattern_length, const float threshold_factor, BackgroundModel*,
                                std::vector<IUPACPattern*>& pats, size_t max_merged_length) {
  struct Similarity {
    float score;
    int shift;
    bool comp;
  };
  std::map<std::pair<size_t, size_t>, Similarity> known;  // exact, by (serial of pats[i], serial of pats[j]), i < j
  std::vector<size_t> serial(pats.size());
  size_t next_serial = 0;
  for (size_t& x : serial) x = next_serial++;
  auto exact = [&](size_t i, size_t j) -> const Similarity& {
    const auto key = std::make_pair(serial[i], serial[j]);
    auto it = known.find(key);
    if (it == known.end()) {
      auto res = IUPACPattern::calculate_S(pats[i], pats[j], strand, bg_model->getV()[0]);
      it = known.emplace(key, Similarity{std::get<0>(res), std::get<1>(res), std::get<2>(res)}).first;
    }
    return it->second;
  };

  // ---- device grid over the eligible motifs (log p <= -5, :253,:256), keyed by serial numbers ---------------------
  size_t n_eligible = 0;
  for (IUPACPattern* p : pats) n_eligible += !(p->getLogPval() > -5);
  const char* force = std::getenv("PENGK_MERGE_GRID");  // "host": exact scores only (tests compare the two paths)
  bool grid = n_eligible >= MERGE_GRID_MIN && !(force && std::string(force) == "host");
  for (IUPACPattern* p : pats) grid = grid && p->get_pattern_length() <= (size_t)PENGK_MAX_MOTIF_LEN;
  const size_t S = 2 * pats.size() + 2;  // serial numbers never exceed n + merges <= 2 n
  std::vector<float> approx;
  const float none = std::numeric_limits<float>::quiet_NaN();
  auto grid_update = [&](size_t first_new_pos) {
    // every pair (i, j), i < j, with pats[j] at position >= first_new_pos, among the eligible motifs
    std::vector<size_t> el;
    for (size_t i = 0; i < pats.size(); ++i)
      if (!(pats[i]->getLogPval() > -5)) el.push_back(i);
    size_t first_new = el.size();
    for (size_t k = 0; k < el.size(); ++k)
      if (el[k] >= first_new_pos) {
        first_new = k;
        break;
      }
    const int n = (int)el.size();
    size_t pairs = 0;
    for (size_t j = first_new; j < el.size(); ++j) pairs += j;
    if (!pairs) return;
    const size_t cell = (size_t)PENGK_MAX_MOTIF_LEN * 4;
    std::vector<float> pw(n * cell, 0.0f), cp(n * cell, 0.0f), out(pairs);
    std::vector<int32_t> len(n);
    std::vector<uint64_t> sites(n);
    for (int k = 0; k < n; ++k) {
      IUPACPattern* p = pats[el[k]];
      len[k] = (int32_t)p->get_pattern_length();
      sites[k] = (uint64_t)p->get_sites();
      for (int c = 0; c < len[k]; ++c)
        for (int a = 0; a < 4; ++a) {
          pw[k * cell + c * 4 + a] = p->get_pwm()[c][a];
          cp[k * cell + c * 4 + a] = p->get_comp_pwm()[c][a];
        }
    }
    pengk_host::check(pengk_motif_similarity(pengk_host::context(), n, pw.data(), cp.data(), len.data(), sites.data(),
                                             strand == Strand::BOTH_STRANDS, bg_model->getV()[0], (int)first_new, out.data()),
                      "pengk_motif_similarity");
    size_t q = 0;
    for (size_t j = first_new; j < el.size(); ++j)
      for (size_t i = 0; i < j; ++i) approx[serial[el[i]] * S + serial[el[j]]] = out[q++];
  };
  if (grid) {
    approx.assign(S * S, none);
    grid_update(0);
  }

  for (;;) {
    float best_score = -std::numeric_limits<float>::infinity();
    size_t bi = 0, bj = 0;
    int best_shift = 0;
    bool best_comp = false;
    float floor = -std::numeric_limits<float>::infinity();  // device scores below it cannot be the maximum
    if (grid) {
      float amax = -std::numeric_limits<float>::infinity();
      size_t longest = 0;
      for (size_t i = 0; i < pats.size(); ++i) {
        if (pats[i]->getLogPval() > -5) continue;
        longest = std::max(longest, (size_t)pats[i]->get_pattern_length());
        const float* row = &approx[serial[i] * S];
        for (size_t j = i + 1; j < pats.size(); ++j)
          if (!(pats[j]->getLogPval() > -5) && row[serial[j]] > amax) amax = row[serial[j]];
      }
      floor = amax - merge_margin(longest);
    }
    for (size_t i = 0; i < pats.size(); ++i) {
      if (pats[i]->getLogPval() > -5) continue;
      for (size_t j = i + 1; j < pats.size(); ++j) {
        if (pats[j]->getLogPval() > -5) continue;
        if (grid && approx[serial[i] * S + serial[j]] < floor) continue;
        const Similarity& sim = exact(i, j);
        if (sim.score > best_score) {
          best_score = sim.score;
          bi = i;
          bj = j;
          best_shift = sim.shift;
          best_comp = sim.comp;
        }
      }
    }
    if (!(best_score > pattern_length * threshold_factor && pats[bi]->get_pattern_length() <= max_merged_length &&
          pats[bj]->get_pattern_length() <= max_merged_length))
      return;
    IUPACPattern* longer = pats[bi];
    IUPACPattern* shorter = pats[bj];
    if (longer->get_pattern_length() < shorter->get_pattern_length()) std::swap(longer, shorter);
    IUPACPattern* merged = new IUPACPattern(longer, shorter, best_comp, bg_model->getV()[0], best_shift);
    const size_t mlen = merged->get_pattern_length();
    if (!(mlen <= sequence_set->getMaxL() && mlen <= max_merged_length)) {
      // the reference keeps looping on the same best pair here (src/peng.cpp:308-310, `continue` with
      // found_better == false ends its while loop): stop merging
      delete merged;
      return;
    }
    std::cout << "merge: " << pats[bj]->get_pattern_string() << " + " << pats[bi]->get_pattern_string() << " -> "
              << merged->get_pattern_string() << std::endl;
    delete pats[bj];
    delete pats[bi];
    pats.erase(pats.begin() + bj);
    pats.erase(pats.begin() + bi);
    pats.push_back(merged);
    serial.erase(serial.begin() + bj);
    serial.erase(serial.begin() + bi);
    serial.push_back(next_serial++);
    if (grid) {
      if (merged->get_pattern_length() > (size_t)PENGK_MAX_MOTIF_LEN) grid = false;  // (max_merged_length > 64: exact scores only)
      else grid_update(pats.size() - 1);
    }
  }
}

// ---- one pass over pattern length W (src/peng.cpp:322-435) ----------------------------------------------------
void Peng::process(PengParameters& params, std::vector<IUPACPattern*>& best_iupac_patterns) {
  const double lim11 = std::log((double)SIZE_MAX) / std::log((double)IUPAC_ALPHABET_SIZE) - 1;
  const double lim4 = std::log((double)SIZE_MAX) / std::log((double)Alphabet::getSize()) - 1;
  if (params.max_pattern_length > lim11 || params.max_pattern_length > lim4) {
    std::cerr << "Warning: pattern length too long!" << std::endl;
    std::cerr << "max pattern length: " << std::max(lim11, lim4) << std::endl;
    exit(1);
  }
  const size_t W = params.max_pattern_length;
  if (W > (size_t)PENGK_MAX_W) {
    // W = 16: the reference passes its own limit (11^16 < 2^64) and then asks for 4^16-entry tables -- 34 GB of counters,
    // ~150 GB in all; its README advises W <= 12 (README.md:119).  This build's tables end at W = 14 (1 GiB each): the
    // same two lines and exit code the reference gives a pattern length beyond ITS limit (src/peng.cpp:325-330).
    std::cerr << "Warning: pattern length too long!" << std::endl;
    std::cerr << "max pattern length: " << PENGK_MAX_W << std::endl;
    exit(1);
  }
  print_status("Processing kmers of length " + std::to_string(W), false);
  print_status("Finding overrepresented kmers (base patterns)", false);
  const int cur_k = std::min((int)W - 1, k), cur_max_k = std::min((int)W - 1, max_k);
  pengk_host::Lap lap;
  BasePattern* base = new BasePattern(W, strand, cur_k, cur_max_k, sequence_set, bg_model);
  lap("base patterns (pack, upload, count, sweep)");

  auto seeds = base->select_base_patterns(params.zscore_threshold, params.count_threshold, strand == Strand::PLUS_STRAND,
                                          params.filter_neighbors);
  if (seeds.empty()) std::cout << "No overrepresented seed patterns found. Stopping." << std::endl;
  lap("seed selection");
  base->print_patterns(seeds);

  print_status("Optimizing base patterns");
  std::cout << std::endl;
  if (seeds.size() > params.max_optimized_patterns) seeds.resize(params.max_optimized_patterns);
  std::vector<IUPACPattern*> unoptimized;
  optimize_iupac_patterns(params.opt_score_type, base, seeds, unoptimized, params.enrich_pseudocount_factor);
  std::cout << std::endl;
  lap("hill-climb");

  print_status("Filtering degenerated IUPAC patterns");
  filter_iupac_patterns(W, params.minimum_processed_motifs, unoptimized);
  for (auto p : unoptimized) std::cout << "selected iupac pattern: " << IUPACPattern::toString(p->get_pattern(), W) << std::endl;

  print_status("Calculating PWMs");
  for (IUPACPattern* p : unoptimized) {
    if (params.adv_pwm) {
      std::cout << "adv pwm: ";
      p->calculate_adv_pwm(base, params.pseudo_counts, nullptr, bg_model->getV()[0]);  // (aggregates on the device)
    } else {
      std::cout << "def pwm: ";
      p->calculate_pwm(base, params.pseudo_counts, base->getPatternCounter(), bg_model->getV()[0]);
    }
    const float avg_info = calculate_pwm_info(p->get_pwm(), (unsigned)W, (unsigned)alphabet_size) / W;
    std::cout << IUPACPattern::toString(p->get_pattern(), W) << " -> " << p->get_pattern_string()
              << "   [ avg. info: " << std::setprecision(2) << avg_info << " ]" << std::endl;
  }

  lap("filter + PWMs");
  print_status("Optimizing expectation-maximization / merging patterns");
  {
    const int background = max_k > (int)W - 1 ? (int)W - 1 : max_k;
    std::cout << std::endl << "background order: " << max_k << std::endl;
    std::vector<IUPACPattern*> optimized;
    if (params.use_em) {
      em_optimize_pwms(unoptimized, base, params.em_saturation_factor, params.em_min_threshold, params.em_max_iterations,
                       background, optimized);
    } else {
      optimized = std::move(unoptimized);
      unoptimized.clear();
    }
    lap("EM");
    if (params.use_merging) {
      if (W >= (size_t)MIN_MERGE_OVERLAP) {
        merge_iupac_patterns(W, params.bit_factor_merge_threshold, bg_model, optimized, params.max_merged_length);
      } else {
        std::cerr << "Warning: Specified pattern length (" << W << ") is too low for merging!" << std::endl;
      }
    }
    lap("merging");
    for (IUPACPattern* p : optimized) {
      p->set_optimization_bg_model_order(max_k);
      best_iupac_patterns.push_back(p);
    }
  }
  for (IUPACPattern* p : unoptimized) delete p;
  delete base;
}

// ---- hill-climb in IUPAC space (src/peng.cpp:437-541) -----------------------------------------------------------
// The reference climbs seed after seed; a climb scores all single-letter mutants of its current mother per round
// and ends when no mutant is better -- or when its best pattern was already met by an EARLIER seed's climb (:504-514).
// Only that second test couples the seeds, and it can only end a climb sooner.  So all climbs advance in lockstep here,
// ignoring it: every round scores the mutants of every still-improving seed in ONE device launch (K4), on the same
// float scores and in the reference's order (position-major, neighbourhood order) within a seed.  Afterwards the
// recorded rounds are replayed seed by seed with the `seen` bookkeeping, cutting each climb where the reference would
// have stopped it; output and result are those of the sequential walk.
void Peng::optimize_iupac_patterns(OPTIMIZATION_SCORE score_type, BasePattern* base_patterns,
                                   std::vector<size_t>& selected_base_patterns, std::vector<IUPACPattern*>& best_iupac_patterns,
                                   float enrich_pseudocount_factor) {
  std::set<size_t> seen, best;
  const size_t W = base_patterns->getPatternLength();
  const size_t pseudo_expected = (size_t)(sequence_set->getN() * enrich_pseudocount_factor);

  struct Round {
    std::vector<std::pair<IUPACPattern*, float>> accepted;  // every improvement of the round, in the order it was found
    std::set<size_t> mutants;                              // all patterns scored in the round
  };
  struct Climb {
    IUPACPattern* start = nullptr;
    float start_score = 0;
    IUPACPattern* mother = nullptr;  // best pattern so far (not owned twice: it is `start` or an accepted mutant)
    float score = 0;
    bool improving = true;
    std::vector<Round> rounds;
  };
  std::vector<Climb> climbs(selected_base_patterns.size());
  {
    std::vector<IUPACPattern*> starts;
    for (size_t s = 0; s < climbs.size(); ++s) {
      climbs[s].start = climbs[s].mother = new IUPACPattern(base_patterns->baseId2IUPACId(selected_base_patterns[s]), W);
      starts.push_back(climbs[s].start);
    }
    IUPACPattern::aggregate_batch(base_patterns, starts);
    for (size_t s = 0; s < climbs.size(); ++s)
      climbs[s].start_score = climbs[s].score = base_patterns->getOptimizationScore(score_type, selected_base_patterns[s], pseudo_expected);
  }
  for (;;) {
    std::vector<IUPACPattern*> mutants;
    std::vector<size_t> first_of(climbs.size() + 1, 0);
    for (size_t s = 0; s < climbs.size(); ++s) {
      first_of[s] = mutants.size();
      if (!climbs[s].improving) continue;
      const size_t mother = climbs[s].mother->get_pattern();
      for (size_t p = 0; p < W; ++p) {
        const int c = IUPACPattern::getNucleotideAtPos(mother, p);
        const size_t masked = mother - c * IUPACPattern::iupac_factor[p];
        for (int r : IUPACAlphabet::similar(c)) mutants.push_back(new IUPACPattern(masked + r * IUPACPattern::iupac_factor[p], W));
      }
    }
    first_of[climbs.size()] = mutants.size();
    if (mutants.empty()) break;
    IUPACPattern::aggregate_batch(base_patterns, mutants);
    for (size_t s = 0; s < climbs.size(); ++s) {
      Climb& c = climbs[s];
      if (!c.improving) continue;
      Round round;
      for (size_t i = first_of[s]; i < first_of[s + 1]; ++i) {
        IUPACPattern* m = mutants[i];
        round.mutants.insert(m->get_pattern());
        const float score = m->getOptimizationScore(score_type, pseudo_expected, (unsigned)n_sequences);
        if (score < c.score) {
          round.accepted.emplace_back(m, score);
          c.mother = m;
          c.score = score;
        } else {
          delete m;
        }
      }
      c.improving = !round.accepted.empty();
      c.rounds.push_back(std::move(round));
    }
  }

  for (size_t s = 0; s < climbs.size(); ++s) {
    Climb& c = climbs[s];
    const size_t seed = selected_base_patterns[s];
    IUPACPattern* best_mutant = c.start;
    print_row(best_mutant, W, c.start_score, true);
    size_t used = 0;
    for (Round& round : c.rounds) {
      ++used;
      for (auto& a : round.accepted) {
        delete best_mutant;
        best_mutant = a.first;
        print_row(best_mutant, W, a.second, true);
      }
      bool found_better = !round.accepted.empty();
      if (seen.count(best_mutant->get_pattern()) == 1) found_better = false;
      round.mutants.erase(best_mutant->get_pattern());
      seen.insert(round.mutants.begin(), round.mutants.end());
      if (!found_better) break;
    }
    for (size_t r = used; r < c.rounds.size(); ++r)  // rounds the reference would not have run
      for (auto& a : c.rounds[r].accepted) delete a.first;

    if (best.count(best_mutant->get_pattern()) == 0 && seen.count(best_mutant->get_pattern()) == 0) {
      best_iupac_patterns.push_back(best_mutant);
      best.insert(best_mutant->get_pattern());
      seen.insert(best_mutant->get_pattern());
      std::cout << "optimization: " << base_patterns->toString(seed) << " -> "
                << IUPACPattern::toString(best_mutant->get_pattern(), W) << std::endl
                << std::endl;
    } else {
      std::cout << "optimization: " << base_patterns->toString(seed) << " removed" << '\t' << std::endl << std::endl;
      delete best_mutant;
    }
  }

  std::cout << std::setw(15) << "pattern" << "\t" << std::setw(15) << "observed" << "\t" << std::setw(15) << "enrichment"
            << "\t" << std::setw(15) << "zscore" << std::endl
            << std::endl;
  std::cout << std::fixed << std::setprecision(2);
  for (auto p : best_iupac_patterns)
    std::cout << std::setw(15) << IUPACPattern::toString(p->get_pattern(), W) << "\t" << std::setw(15) << p->get_sites() << "\t"
              << std::setw(15) << (p->get_sites() / p->getExpectedCounts()) << "\t" << std::setw(15) << p->getZscore()
              << std::endl;
}

// ---- drop near-all-N patterns and weak ones (src/peng.cpp:543-599) ------------------------------------------------
void Peng::filter_iupac_patterns(size_t pattern_length, size_t minimum_retained_motifs, std::vector<IUPACPattern*>& pats) {
  std::vector<IUPACPattern*> kept, dropped;
  for (IUPACPattern* p : pats) {
    size_t n_wild = 0;
    for (size_t q = 0; q < pattern_length; ++q)
      if (IUPACPattern::getNucleotideAtPos(p->get_pattern(), q) == (int)IUPAC_Alphabet::N) ++n_wild;
    (pattern_length - n_wild <= 3 ? dropped : kept).push_back(p);
  }
  std::sort(kept.begin(), kept.end(), sort_IUPAC_patterns);
  float min_pvalue = -5.f;
  if (!kept.empty()) min_pvalue = std::min(-5.f, kept[0]->getLogPval() * 0.2f);
  pats.clear();
  for (size_t i = 0; i < kept.size(); ++i) {
    if (kept[i]->getLogPval() < min_pvalue || i < minimum_retained_motifs) pats.push_back(kept[i]);
    else dropped.push_back(kept[i]);
  }
  for (IUPACPattern* p : dropped) delete p;
}

// ---- writers (src/peng.cpp:602-728) ----------------------------------------------------------------------------------
void Peng::printShortMeme(std::vector<IUPACPattern*>& pats, const std::string output_filename, BackgroundModel* bg) {
  const unsigned PRECISION = 8;
  std::sort(pats.begin(), pats.end(), sort_IUPAC_patterns);
  std::ofstream out(output_filename);
  if (!out.is_open()) {
    std::cerr << "Unable to open output file (" << output_filename << ")!";
    return;
  }
  const char* alphabet = Alphabet::getAlphabet();
  out << "MEME version 4" << std::endl << std::endl;
  out << "ALPHABET= " << alphabet << std::endl << std::endl;
  out << "Background letter frequencies" << std::endl;
  for (size_t i = 0; i < strlen(alphabet); ++i) out << (i ? " " : "") << alphabet[i] << " " << bg->getV()[0][i];
  out << std::endl << std::endl;
  for (IUPACPattern* p : pats) {
    out << "MOTIF " << p->get_pattern_string() << std::endl;
    out << "letter-probability matrix:" << " alength= " << 4 << " w= " << p->get_pattern_length() << " nsites= " << p->get_sites()
        << " bg_prob= " << p->get_bg_p() << " opt_bg_order= " << p->get_optimization_bg_model_order()
        << " log(Pval)= " << p->getLogPval() << std::endl;
    float** pwm = p->get_pwm();
    no_zero_pwm(pwm, (unsigned)p->get_pattern_length(), PRECISION);
    for (size_t w = 0; w < p->get_pattern_length(); ++w) {
      for (size_t a = 0; a < 4; ++a) out << (a ? " " : "") << std::fixed << std::setprecision(PRECISION) << pwm[w][a];
      out << std::endl;
    }
    out << std::endl;
  }
}

void Peng::printJson(std::vector<IUPACPattern*>& pats, const std::string output_filename, const std::string,
                     BackgroundModel* bg) {
  const unsigned PRECISION = 8;
  std::sort(pats.begin(), pats.end(), sort_IUPAC_patterns);
  std::ofstream out(output_filename);
  if (!out.is_open()) {
    std::cerr << "Unable to open output file (" << output_filename << ")!";
    return;
  }
  const char* alphabet = Alphabet::getAlphabet();
  out << "{" << std::endl;
  out << "\t\"alphabet\" : \"" << alphabet << "\"," << std::endl;
  out << "\t\"bg\" : [";
  for (size_t i = 0; i < strlen(alphabet); ++i) out << bg->getV()[0][i] << (i + 1 != strlen(alphabet) ? ", " : "");
  out << "]," << std::endl;
  out << "\t\"alphabet_length\" : " << 4 << "," << std::endl;
  out << "\t\"patterns\" : [" << std::endl;
  for (size_t n = 0; n < pats.size(); ++n) {
    IUPACPattern* p = pats[n];
    out << "\t\t{" << std::endl;
    out << "\t\t\t\"iupac_motif\" : \"" << p->get_pattern_string() << "\"," << std::endl;
    out << "\t\t\t\"pattern_length\" : " << p->get_pattern_length() << "," << std::endl;
    out << "\t\t\t\"sites\" : " << p->get_sites() << "," << std::endl;
    out << "\t\t\t\"log(Pval)\" : " << p->getLogPval() << "," << std::endl;
    out << "\t\t\t\"bg_prob\" : " << p->get_bg_p() << "," << std::endl;
    out << "\t\t\t\"opt_bg_order\" : " << p->get_optimization_bg_model_order() << "," << std::endl;
    out << "\t\t\t\"pwm\" : [" << std::endl;
    float** pwm = p->get_pwm();
    no_zero_pwm(pwm, (unsigned)p->get_pattern_length(), PRECISION);
    for (size_t w = 0; w < p->get_pattern_length(); ++w) {
      out << "\t\t\t\t\t[";
      for (size_t a = 0; a < 4; ++a) out << std::fixed << std::setprecision(PRECISION) << pwm[w][a] << (a != 3 ? ", " : "]");
      if (w + 1 != p->get_pattern_length()) out << ", ";
      out << std::endl;
    }
    out << "\t\t\t\t]" << std::endl;
    out << "\t\t}" << (n + 1 != pats.size() ? "," : "") << std::endl;
  }
  out << "\t]" << std::endl;
  out << "}" << std::endl;
}
