// IUPACPattern -- a degenerate (IUPAC) pattern with its statistics and PWM.  Public surface of the
// reference's src/iupac_pattern.h; the aggregation over the underlying k-mers
// (aggregate_attributes_from_basepatterns, count_combined_occurences: src/iupac_pattern.cpp:410-473,
// 806-833) runs on the GPU through pengk_iupac_aggregate, single patterns or whole hill-climb rounds.
//
// ids: little-endian base 11, letters A C G T S W R Y M K N = 0..10.
#ifndef PENGK_HOST_IUPAC_PATTERN_H_
#define PENGK_HOST_IUPAC_PATTERN_H_

#include <cstdlib>
#include <string>
#include <tuple>
#include <vector>

#include "Global.h"

class BasePattern;

static const int MIN_MERGE_OVERLAP = 6;

class IUPACPattern {
 public:
  static size_t* iupac_factor;  // 11^i

  static void init(size_t max_pattern_length, float* bg_model);
  static void initIUPACProfile(const float mixin_factor, const float mixin_bias, float* bg_model);
  static std::string toString(size_t pattern_id, size_t pattern_length);
  static int getNucleotideAtPos(const size_t pattern, const size_t pos);
  static void normalize_pwm(const int pattern_length, float** pwm);
  static float calculate_s(float** p1_pwm, float** p2_pwm, float* background, const int offset1, const int offset2,
                           const int l);
  static std::tuple<float, int, bool> calculate_S(IUPACPattern* p1, IUPACPattern* p2, Strand s, float* background);

  IUPACPattern(size_t iupac_pattern, size_t pattern_length);
  IUPACPattern(IUPACPattern* ori, float** pwm);
  IUPACPattern(IUPACPattern* longer_pattern, IUPACPattern* shorter_pattern, bool is_comp, float* background,
               const int shift);
  ~IUPACPattern();

  size_t get_pattern() { return pattern; }
  std::string get_pattern_string();
  size_t get_pattern_length() { return pattern_length; }

  float getExpCountFraction(const size_t pseudo_expected_pattern_counts);
  float getLogPval() { return log_pvalue; }
  float getMutualInformationScore(unsigned int n_sequences);
  float getOptimizationScore(OPTIMIZATION_SCORE score_type, const size_t pseudo_expected_pattern_counts,
                             unsigned int n_sequences);
  float get_bg_p() { return bg_p; }
  float** get_pwm() { return pwm; }
  float** get_comp_pwm() { return comp_pwm; }
  void calculate_comp_pwm();
  void update_pwm(float** new_pwm);
  size_t get_sites() { return n_sites; }
  size_t* get_local_sites() { return local_n_sites; }
  float getExpectedCounts() const { return expected_counts; }
  float getZscore() const { return zscore; }
  std::vector<size_t>& get_base_patterns() { return base_patterns; }
  int get_optimization_bg_model_order() { return optimization_bg_model_order; }
  void set_optimization_bg_model_order(int order) { optimization_bg_model_order = order; }

  void calculate_pwm(BasePattern* base_pattern, const int pseudo_counts, size_t* pattern_counter, float* background_model);
  void calculate_adv_pwm(BasePattern* base_pattern, const int pseudo_counts, size_t* pattern_counter,
                         float* background_model);
  void aggregate_attributes_from_basepatterns(BasePattern*);
  // one device launch for a whole set of patterns (a hill-climb round)
  static void aggregate_batch(BasePattern*, const std::vector<IUPACPattern*>& patterns);

  bool operator<(const IUPACPattern& rhs) const { return pattern < rhs.pattern; }

  std::vector<size_t> generate_base_patterns(BasePattern* basepatterns, size_t iupac_pattern);
  unsigned long count_combined_occurences(BasePattern*, size_t iupac_pattern);
  std::vector<size_t> basepatterns_from_iupac_single_stranded(BasePattern*, size_t iupac_pattern);
  std::vector<size_t> basepatterns_from_iupac_double_stranded(BasePattern*, size_t iupac_pattern);
  static void find_base_patterns(BasePattern* base_pattern, const size_t pattern, const size_t pattern_length,
                                 std::vector<size_t>& base_patterns);

 private:
  static float calculate_d(float** p1_pwm, float** p2_pwm, const int offset1, const int offset2, const int l,
                           const float epsilon = 1E-4);
  static float calculate_d_bg(float** p_pwm, float* background, const int l, const int offset = 0,
                              const float epsilon = 1E-4);
  float calculate_merged_pvalue(IUPACPattern* longer_pattern, IUPACPattern* shorter_pattern, bool is_comp,
                                float* background, const int shift);
  void alloc_pwm();

  static float** iupac_profile;
  static float* log_bonferroni;

  size_t pattern_length;
  size_t pattern;
  float log_pvalue;
  float zscore;
  float bg_p;
  float expected_counts;
  int optimization_bg_model_order;
  size_t n_sites;
  size_t* local_n_sites;
  float** pwm;
  float** comp_pwm;
  std::vector<size_t> base_patterns;
  bool merged;
};

bool sort_IUPAC_patterns(IUPACPattern* a, IUPACPattern* b);

#endif
