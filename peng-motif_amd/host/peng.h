// Peng -- orchestration of one PEnG-motif run (public surface of the reference's src/peng.h).
// Control flow (seed selection, hill-climb decisions, filtering, merging, writers) is host code over
// <= 50 seeds; the scoring inside it runs on the device: each hill-climb round is ONE
// pengk_iupac_aggregate launch over all mutants, the EM of all PWMs is ONE pengk_em call.
#ifndef PENGK_HOST_PENG_H_
#define PENGK_HOST_PENG_H_

#include <array>
#include <set>
#include <string>
#include <tuple>
#include <vector>

#include "Global.h"
#include "iupac_pattern.h"
#include "shared/Alphabet.h"
#include "shared/BackgroundModel.h"
#include "shared/SequenceSet.h"

class PengParameters {
 public:
  size_t max_pattern_length;
  float zscore_threshold;
  size_t count_threshold;
  int pseudo_counts;
  OPTIMIZATION_SCORE opt_score_type;
  float enrich_pseudocount_factor;
  bool use_em;
  float em_saturation_factor;
  float em_min_threshold;
  int em_max_iterations;
  bool use_merging;
  float bit_factor_merge_threshold;
  bool adv_pwm;
  size_t minimum_processed_motifs;
  bool filter_neighbors;
  size_t max_optimized_patterns;
  size_t max_merged_length;
};

class Peng {
 public:
  Peng(Strand s, const int k, const int max_opt_k, SequenceSet* sequence_set, BackgroundModel* bg);
  ~Peng();

  void process(PengParameters& params, std::vector<IUPACPattern*>& best_iupac_patterns);
  void filter_redundancy(const float merge_bit_factor_threshold, std::vector<IUPACPattern*>& iupac_patterns);
  void printShortMeme(std::vector<IUPACPattern*>& best_iupac_patterns, const std::string output_filename,
                      BackgroundModel* bg_model);
  void printJson(std::vector<IUPACPattern*>& best_iupac_patterns, const std::string output_filename,
                 const std::string version_number, BackgroundModel* bg_model);

 private:
  BackgroundModel* bg_model;
  SequenceSet* sequence_set;
  int max_k;
  int k;
  int alphabet_size;
  size_t n_sequences;
  Strand strand;

  void optimize_iupac_patterns(OPTIMIZATION_SCORE score_type, BasePattern* base_patterns,
                               std::vector<size_t>& selected_base_patterns, std::vector<IUPACPattern*>& best_iupac_patterns,
                               float enrich_pseudocount_factor);
  void filter_iupac_patterns(size_t pattern_length, size_t minimum_retained_motifs, std::vector<IUPACPattern*>& iupac_patterns);
  void merge_iupac_patterns(size_t pattern_length, float bit_factor_merge_threshold, BackgroundModel* bg,
                            std::vector<IUPACPattern*>& iupac_patterns, size_t max_merged_length);
  void em_optimize_pwms(std::vector<IUPACPattern*>& iupac_patterns, BasePattern* base_patterns, float saturation_factor,
                        float min_em_threshold, int max_iterations, int background_order,
                        std::vector<IUPACPattern*>& optimized_iupac_patterns);
};

#endif
