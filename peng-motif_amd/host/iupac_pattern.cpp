// IUPACPattern -- host mirror.  Aggregation over the underlying k-mers is a device call; the PWM
// algebra (similarity, merging, naming) is O(#motifs * W) scalar arithmetic and stays on the host,
// written to round exactly like the reference (float variables, log2 in double).
#include "iupac_pattern.h"

#include <algorithm>
#include <cmath>
#include <iostream>
#include <limits>

#include "base_pattern.h"
#include "device.h"
#include "iupac_alphabet.h"
#include "shared/Alphabet.h"
#include "utils.h"

size_t* IUPACPattern::iupac_factor = nullptr;
float* IUPACPattern::log_bonferroni = nullptr;
float** IUPACPattern::iupac_profile = nullptr;

namespace {
constexpr float kMixinFactor = 0.2;
constexpr float kMixinBias = 0.7;

float** new_matrix(size_t rows) {
  float** m = new float*[rows];
  for (size_t p = 0; p < rows; ++p) {
    m[p] = new float[4];
    for (int a = 0; a < 4; ++a) m[p][a] = 0;
  }
  return m;
}
void delete_matrix(float** m, size_t rows) {
  if (!m) return;
  for (size_t p = 0; p < rows; ++p) delete[] m[p];
  delete[] m;
}
inline double lg2(float x) { return std::log2((double)x); }
}  // namespace

// ---- statics ---------------------------------------------------------------------------------------
void IUPACPattern::init(size_t max_pattern_length, float* bg_model) {
  delete[] iupac_factor;
  iupac_factor = new size_t[max_pattern_length + 1];
  for (size_t i = 0; i <= max_pattern_length; ++i) iupac_factor[i] = (size_t)std::pow((double)IUPAC_ALPHABET_SIZE, (double)i);
  delete[] log_bonferroni;
  log_bonferroni = new float[IUPAC_ALPHABET_SIZE];
  for (int c = 0; c < IUPAC_ALPHABET_SIZE; ++c) {
    // number of IUPAC letters a position could have been mutated from/to: ln 8, 16, 24, 6
    const double n = c < 4 ? 8 : (c < 8 ? 16 : (c < 10 ? 24 : 6));
    log_bonferroni[c] = (float)std::log(n);
  }
  initIUPACProfile(kMixinFactor, kMixinBias, bg_model);
}

void IUPACPattern::initIUPACProfile(const float mixin_factor, const float mixin_bias, float* bg_model) {
  if (!iupac_profile) iupac_profile = new_matrix(IUPAC_ALPHABET_SIZE);
  for (int c = 0; c < IUPAC_ALPHABET_SIZE; ++c) {
    const std::vector<int>& rep = IUPACAlphabet::representative(c);
    for (int a = 0; a < 4; ++a) {
      float v = 0;
      v += mixin_factor * bg_model[a];
      if (std::find(rep.begin(), rep.end(), a) != rep.end()) v += mixin_bias;
      iupac_profile[c][a] = v;
    }
  }
}

std::string IUPACPattern::toString(size_t pattern_id, size_t pattern_length) {
  std::string out;
  for (size_t p = 0; p < pattern_length; ++p) out += IUPACAlphabet::getBase(getNucleotideAtPos(pattern_id, p));
  return out;
}

int IUPACPattern::getNucleotideAtPos(const size_t pattern, const size_t pos) {
  return (int)((pattern / iupac_factor[pos]) % IUPAC_ALPHABET_SIZE);
}

void IUPACPattern::normalize_pwm(const int pattern_length, float** pwm) {
  for (int p = 0; p < pattern_length; ++p) {
    float sum = 0;
    for (int a = 0; a < 4; ++a) sum += pwm[p][a];
    for (int a = 0; a < 4; ++a) pwm[p][a] /= sum;
  }
}

// Jensen-Shannon-like distance between two PWM stretches (float accumulator, log2 in double)
float IUPACPattern::calculate_d(float** p1_pwm, float** p2_pwm, const int offset1, const int offset2, const int l,
                                const float epsilon) {
  float d = 0;
  for (int i = 0; i < l; ++i)
    for (int a = 0; a < 4; ++a) {
      const float x = p1_pwm[offset1 + i][a], y = p2_pwm[offset2 + i][a];
      const float mean = (x + y + 2 * epsilon) / 2;
      d = (float)((double)d + ((double)(x + epsilon) * lg2(x + epsilon) + (double)(y + epsilon) * lg2(y + epsilon) -
                               (double)(2 * mean) * lg2(mean)));
    }
  return d;
}

float IUPACPattern::calculate_d_bg(float** p_pwm, float* background, const int l, const int offset, const float epsilon) {
  float d = 0;
  for (int i = 0; i < l; ++i)
    for (int a = 0; a < 4; ++a) {
      const float x = p_pwm[offset + i][a], y = background[a];
      const float mean = (x + y + 2 * epsilon) / 2;
      d = (float)((double)d + ((double)(x + epsilon) * lg2(x + epsilon) + (double)(y + epsilon) * lg2(y + epsilon) -
                               (double)(2 * mean) * lg2(mean)));
    }
  return d;
}

float IUPACPattern::calculate_s(float** p1_pwm, float** p2_pwm, float* background, const int offset1, const int offset2,
                                const int l) {
  const float both_bg = calculate_d_bg(p1_pwm, background, l, offset1) + calculate_d_bg(p2_pwm, background, l, offset2);
  return (float)(0.5 * (double)both_bg - (double)calculate_d(p1_pwm, p2_pwm, offset1, offset2, l));
}

// best similarity over all shifts with >= MIN_MERGE_OVERLAP overlapping columns (and the reverse
// complement when both strands are searched): (score, shift, used the complement)
std::tuple<float, int, bool> IUPACPattern::calculate_S(IUPACPattern* p1, IUPACPattern* p2, Strand s, float* background) {
  IUPACPattern* big = p1;
  IUPACPattern* small = p2;
  if (p1->get_pattern_length() < p2->get_pattern_length()) std::swap(big, small);
  const int lb = (int)big->get_pattern_length(), ls = (int)small->get_pattern_length();
  float best = -std::numeric_limits<float>::infinity();
  int best_shift = -255;
  bool best_comp = false;
  const int n_orient = s == Strand::BOTH_STRANDS ? 2 : 1;
  for (int orient = 0; orient < n_orient; ++orient) {
    const bool comp = orient == 1;
    float** pb = big->get_pwm();
    float** ps = small->get_pwm();
    if (comp) {
      if (big->get_sites() < small->get_sites()) pb = big->get_comp_pwm();
      else ps = small->get_comp_pwm();
    }
    for (int shift = MIN_MERGE_OVERLAP - ls; shift <= lb - MIN_MERGE_OVERLAP; ++shift) {
      const int off_small = -std::min(shift, 0), off_big = std::max(shift, 0);
      const int overlap = std::min(lb - off_big, ls - off_small);
      const float sc = calculate_s(pb, ps, background, off_big, off_small, overlap);
      if (sc > best) {
        best = sc;
        best_shift = shift;
        best_comp = comp;
      }
    }
  }
  return std::make_tuple(best, best_shift, best_comp);
}

// ---- construction ----------------------------------------------------------------------------------
IUPACPattern::IUPACPattern(size_t iupac_pattern, size_t pattern_length)
    : pattern_length(pattern_length), pattern(iupac_pattern), log_pvalue(0), zscore(0), bg_p(0), expected_counts(0),
      optimization_bg_model_order(0), n_sites(0), local_n_sites(new size_t[pattern_length]()), pwm(nullptr),
      comp_pwm(nullptr), merged(false) {}

IUPACPattern::IUPACPattern(IUPACPattern* ori, float** new_pwm)
    : pattern_length(ori->pattern_length), pattern(ori->pattern), log_pvalue(ori->log_pvalue), zscore(0), bg_p(ori->bg_p),
      expected_counts(ori->expected_counts), optimization_bg_model_order(ori->optimization_bg_model_order),
      n_sites(ori->n_sites), local_n_sites(new size_t[ori->pattern_length]), pwm(nullptr), comp_pwm(nullptr),
      merged(ori->merged) {
  for (size_t i = 0; i < pattern_length; ++i) local_n_sites[i] = ori->local_n_sites[i];
  pwm = new_matrix(pattern_length);
  for (size_t p = 0; p < pattern_length; ++p)
    for (int a = 0; a < 4; ++a) pwm[p][a] = new_pwm[p][a];
  normalize_pwm((int)pattern_length, pwm);
  calculate_comp_pwm();
}

// merge of two overlapping motifs: `shift` = position of the shorter one relative to the longer one
IUPACPattern::IUPACPattern(IUPACPattern* longer_pattern, IUPACPattern* shorter_pattern, bool is_comp, float* background,
                           const int shift)
    : pattern(0), log_pvalue(0), zscore(0), bg_p(0), expected_counts(0), optimization_bg_model_order(0), n_sites(0),
      pwm(nullptr), comp_pwm(nullptr), merged(true) {
  const int ll = (int)longer_pattern->get_pattern_length(), ls = (int)shorter_pattern->get_pattern_length();
  const int start_short = std::max(shift, 0);   // column of the merged motif where the shorter one begins
  const int start_long = -std::min(shift, 0);   // ... and where the longer one begins
  const int overlap = std::min(ll - start_short, ls - start_long);
  float** lp = longer_pattern->get_pwm();
  float** sp = shorter_pattern->get_pwm();
  if (is_comp && longer_pattern->get_sites() < shorter_pattern->get_sites()) lp = longer_pattern->get_comp_pwm();
  else if (is_comp) sp = shorter_pattern->get_comp_pwm();

  pattern_length = (size_t)(ll + ls - overlap);
  local_n_sites = new size_t[pattern_length]();
  for (int p = 0; p < ls; ++p) local_n_sites[start_short + p] += shorter_pattern->local_n_sites[p];
  for (int p = 0; p < ll; ++p) local_n_sites[start_long + p] += longer_pattern->local_n_sites[p];
  int sum_sites = 0;
  for (size_t p = 0; p < pattern_length; ++p) sum_sites += (int)local_n_sites[p];
  n_sites = size_t(sum_sites / pattern_length);

  pwm = new_matrix(pattern_length);
  for (int p = 0; p < (int)pattern_length; ++p) {
    const int ps = p - start_short, pl = p - start_long;
    const bool in_s = ps >= 0 && ps < ls, in_l = pl >= 0 && pl < ll;
    for (int a = 0; a < 4; ++a) {
      if (in_s && in_l) {
        const size_t ws = shorter_pattern->local_n_sites[ps], wl = longer_pattern->local_n_sites[pl];
        pwm[p][a] = ((float)ws * sp[ps][a] + (float)wl * lp[pl][a]) / (float)(ws + wl);
      } else if (in_l) {
        pwm[p][a] = lp[pl][a];
      } else if (in_s) {
        pwm[p][a] = sp[ps][a];
      }
    }
  }
  normalize_pwm((int)pattern_length, pwm);
  calculate_comp_pwm();
  log_pvalue = calculate_merged_pvalue(longer_pattern, shorter_pattern, is_comp, background, shift);
}

IUPACPattern::~IUPACPattern() {
  delete_matrix(pwm, pattern_length);
  delete_matrix(comp_pwm, pattern_length);
  delete[] local_n_sites;
}

// log p-value of a merged motif: the better motif's value plus the other one's scaled by the share of
// its information that lies outside the overlap.  (As in the reference, src/iupac_pattern.cpp:245-250,
// the shorter motif's complement PWM is taken whenever the longer one's is not -- also for same-strand
// merges.)
float IUPACPattern::calculate_merged_pvalue(IUPACPattern* longer_pattern, IUPACPattern* shorter_pattern, bool is_comp,
                                            float* background, const int shift) {
  float** lp = longer_pattern->get_pwm();
  float** sp = shorter_pattern->get_pwm();
  if (is_comp && longer_pattern->get_sites() < shorter_pattern->get_sites()) lp = longer_pattern->get_comp_pwm();
  else sp = shorter_pattern->get_comp_pwm();
  const int ll = (int)longer_pattern->get_pattern_length(), ls = (int)shorter_pattern->get_pattern_length();
  const int off_short = -std::min(shift, 0), off_long = std::max(shift, 0);
  const int overlap = std::min(ll - off_long, ls - off_short);
  const bool longer_better = longer_pattern->log_pvalue < shorter_pattern->log_pvalue;
  IUPACPattern* weak = longer_better ? shorter_pattern : longer_pattern;
  IUPACPattern* strong = longer_better ? longer_pattern : shorter_pattern;
  float** wp = longer_better ? sp : lp;
  const int wlen = longer_better ? ls : ll, woff = longer_better ? off_short : off_long;
  float d;
  if (woff != 0) {  // the weaker motif sticks out on the left
    d = calculate_d_bg(wp, background, woff, 0);
  } else {  // ... or on the right
    const int start = woff + overlap;
    d = calculate_d_bg(wp, background, wlen - start, start);
  }
  const float d_all = calculate_d_bg(wp, background, wlen);
  return strong->getLogPval() + d / d_all * weak->getLogPval();
}

void IUPACPattern::calculate_comp_pwm() {
  if (!comp_pwm) comp_pwm = new_matrix(pattern_length);
  for (size_t p = 0; p < pattern_length; ++p)
    for (int a = 0; a < 4; ++a) comp_pwm[p][a] = pwm[pattern_length - 1 - p][3 - a];
}

void IUPACPattern::update_pwm(float** new_pwm) {
  for (size_t p = 0; p < pattern_length; ++p)
    for (int a = 0; a < 4; ++a) pwm[p][a] = new_pwm[p][a];
  normalize_pwm((int)pattern_length, pwm);
  calculate_comp_pwm();
}

// ---- scores ----------------------------------------------------------------------------------------
float IUPACPattern::getExpCountFraction(const size_t pseudo_expected_pattern_counts) {
  return (expected_counts + (float)pseudo_expected_pattern_counts) / (float)n_sites;
}

float IUPACPattern::getMutualInformationScore(unsigned int n_sequences) {
  return mutual_information_score((float)n_sites, expected_counts, n_sequences);
}

float IUPACPattern::getOptimizationScore(OPTIMIZATION_SCORE score_type, const size_t pseudo_expected_pattern_counts,
                                         const unsigned int n_sequences) {
  switch (score_type) {
    case OPTIMIZATION_SCORE::kLogPval: return getLogPval();
    case OPTIMIZATION_SCORE::kExpCounts: return getExpCountFraction(pseudo_expected_pattern_counts);
    case OPTIMIZATION_SCORE::MutualInformation: return getMutualInformationScore(n_sequences);
  }
  std::cerr << "Error: unknown score type!" << std::endl;
  exit(1);
}

// nearest IUPAC letter per PWM row
std::string IUPACPattern::get_pattern_string() {
  std::string res;
  for (size_t i = 0; i < pattern_length; ++i) {
    float min_dist = std::numeric_limits<float>::infinity();
    int min_iupac = 0;
    for (int m = 0; m < IUPAC_ALPHABET_SIZE; ++m) {
      const float dist = calculate_d(pwm, iupac_profile, (int)i, m, 1, 1E-7);
      if (dist < min_dist) {
        min_dist = dist;
        min_iupac = m;
      }
    }
    res += IUPACAlphabet::getBase(min_iupac);
  }
  return res;
}

// ---- device-backed aggregation ---------------------------------------------------------------------
void IUPACPattern::aggregate_batch(BasePattern* bp, const std::vector<IUPACPattern*>& patterns) {
  std::vector<uint64_t> ids;
  std::vector<IUPACPattern*> todo;
  for (IUPACPattern* p : patterns)
    if (!p->merged) {
      ids.push_back(p->pattern);
      todo.push_back(p);
    }
  if (ids.empty()) return;
  std::vector<pengk_iupac_stats> st(ids.size());
  pengk_host::check(pengk_iupac_aggregate(pengk_host::context(), (int)bp->getPatternLength(),
                                          bp->getStrand() == Strand::BOTH_STRANDS, ids.data(), (int64_t)ids.size(),
                                          bp->device_counts(), bp->device_bgprob(bp->getBackgroundOrder()),
                                          bp->device_expected(), st.data()),
                    "pengk_iupac_aggregate");
  for (size_t i = 0; i < todo.size(); ++i) {
    IUPACPattern* p = todo[i];
    p->bg_p = st[i].bg_p;
    p->expected_counts = st[i].expected;
    p->zscore = st[i].zscore;
    p->n_sites = st[i].sites;
    p->log_pvalue = st[i].log_pvalue;
    for (size_t q = 0; q < p->pattern_length; ++q) p->local_n_sites[q] = st[i].sites;
  }
}

void IUPACPattern::aggregate_attributes_from_basepatterns(BasePattern* base_patterns) {
  aggregate_batch(base_patterns, std::vector<IUPACPattern*>{this});
}

unsigned long IUPACPattern::count_combined_occurences(BasePattern* bp, size_t iupac_pattern) {
  uint64_t id = iupac_pattern;
  pengk_iupac_stats st;
  pengk_host::check(pengk_iupac_aggregate(pengk_host::context(), (int)bp->getPatternLength(),
                                          bp->getStrand() == Strand::BOTH_STRANDS, &id, 1, bp->device_counts(),
                                          bp->device_bgprob(bp->getBackgroundOrder()), bp->device_expected(), &st),
                    "pengk_iupac_aggregate");
  return st.sites;
}

void IUPACPattern::alloc_pwm() { pwm = new_matrix(pattern_length); }

// simple PWM (--use-default-pwm).  `base_patterns` is only filled by find_base_patterns, which the
// pipeline never calls, so this is pseudo counts over (sites + pseudo counts) as in the reference.
void IUPACPattern::calculate_pwm(BasePattern* base_pattern, const int pseudo_counts, size_t* pattern_counter,
                                 float* background_model) {
  if (pwm || merged) return;
  alloc_pwm();
  for (size_t p = 0; p < pattern_length; ++p)
    for (int a = 0; a < 4; ++a) pwm[p][a] = (float)pseudo_counts * background_model[a];
  for (size_t base : base_patterns) {
    const size_t count = pattern_counter[base];
    for (size_t p = 0; p < pattern_length; ++p) pwm[p][base_pattern->getFastNucleotideAtPos(base, p)] += (float)count;
  }
  for (size_t p = 0; p < pattern_length; ++p)
    for (int a = 0; a < 4; ++a) pwm[p][a] = (float)((double)pwm[p][a] / (1.0 * (double)n_sites + (double)pseudo_counts));
  calculate_comp_pwm();
}

// advanced PWM: column p, base i = occurrences of the pattern with letter p replaced by i (one device
// launch for all 4W variants) plus truncated pseudo counts
void IUPACPattern::calculate_adv_pwm(BasePattern* bp, const int pseudo_counts, size_t*, float* background_model) {
  if (pwm || merged) return;
  alloc_pwm();
  std::vector<uint64_t> ids;
  for (size_t p = 0; p < pattern_length; ++p) {
    const int c = getNucleotideAtPos(pattern, p);
    for (int i = 0; i < 4; ++i) ids.push_back(pattern - c * iupac_factor[p] + i * iupac_factor[p]);
  }
  std::vector<pengk_iupac_stats> st(ids.size());
  pengk_host::check(pengk_iupac_aggregate(pengk_host::context(), (int)bp->getPatternLength(),
                                          bp->getStrand() == Strand::BOTH_STRANDS, ids.data(), (int64_t)ids.size(),
                                          bp->device_counts(), bp->device_bgprob(bp->getBackgroundOrder()),
                                          bp->device_expected(), st.data()),
                    "pengk_iupac_aggregate");
  for (size_t p = 0; p < pattern_length; ++p) {
    size_t n_total = 0, i_total[4];
    for (int i = 0; i < 4; ++i) {
      i_total[i] = (size_t)((float)pseudo_counts * background_model[i]);  // truncation, as in the reference
      i_total[i] += st[p * 4 + i].sites;
      n_total += i_total[i];
    }
    for (int i = 0; i < 4; ++i) pwm[p][i] = (float)(1.0 * (double)i_total[i] / (double)n_total);
  }
  calculate_comp_pwm();
}

// ---- host-side expansions (API compatibility; the pipeline aggregates on the device) ----------------
std::vector<size_t> IUPACPattern::generate_base_patterns(BasePattern* bp, size_t iupac_pattern) {
  // depth-first with an explicit stack: last degenerate position varies fastest, rep[0] first then rep[n-1]..rep[1]
  const size_t W = bp->getPatternLength();
  std::vector<size_t> out;
  std::vector<std::pair<size_t, size_t>> stack{{0, 0}};
  while (!stack.empty()) {
    size_t kmer = stack.back().first, pos = stack.back().second;
    stack.pop_back();
    for (; pos < W; ++pos) {
      const std::vector<int>& rep = IUPACAlphabet::representative(getNucleotideAtPos(iupac_pattern, pos));
      for (size_t j = 1; j < rep.size(); ++j) stack.push_back({bp->add_letter_to_the_right(kmer, pos, rep[j]), pos + 1});
      kmer = bp->add_letter_to_the_right(kmer, pos, rep[0]);
    }
    out.push_back(kmer);
  }
  return out;
}

std::vector<size_t> IUPACPattern::basepatterns_from_iupac_single_stranded(BasePattern* bp, size_t iupac_pattern) {
  return generate_base_patterns(bp, iupac_pattern);
}

std::vector<size_t> IUPACPattern::basepatterns_from_iupac_double_stranded(BasePattern* bp, size_t iupac_pattern) {
  std::vector<size_t> ids = generate_base_patterns(bp, iupac_pattern);
  for (size_t& x : ids) x = std::min(x, bp->getFastRevCompId(x));
  std::sort(ids.begin(), ids.end());
  return ids;
}

void IUPACPattern::find_base_patterns(BasePattern* base_pattern, const size_t pattern, const size_t pattern_length,
                                      std::vector<size_t>& base_patterns) {
  std::vector<size_t> ids{0};
  const size_t* f = base_pattern->getFactors();
  for (size_t p = 0; p < pattern_length; ++p) {
    std::vector<size_t> next;
    for (int r : IUPACAlphabet::representative(getNucleotideAtPos(pattern, p)))
      for (size_t id : ids) next.push_back(id + (size_t)r * f[p]);
    ids.swap(next);
  }
  base_patterns.insert(base_patterns.end(), ids.begin(), ids.end());
}

bool sort_IUPAC_patterns(IUPACPattern* a, IUPACPattern* b) { return a->getLogPval() < b->getLogPval(); }
