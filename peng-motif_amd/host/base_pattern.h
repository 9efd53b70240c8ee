// BasePattern -- all 4^W base patterns of one length: counts, background probabilities, expected
// counts, log-p and z-scores.  Public surface of the reference's src/base_pattern.h (borrowed raw host
// pointers with the same element types); the tables are produced by the HIP kernels behind
// include/pengk.h; a table is mirrored into a (page-locked) host array the first time it is asked for, so a
// run pays the device-to-host copy only for the tables it reads (the pipeline: counts, expected, z).
//
// ids: little-endian base 4, A C G T = 0..3 ("ATGC" = 0 + 3*4 + 2*16 + 1*64).
#ifndef PENGK_HOST_BASE_PATTERN_H_
#define PENGK_HOST_BASE_PATTERN_H_

#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "Global.h"
#include "device.h"
#include "iupac_pattern.h"
#include "shared/Alphabet.h"
#include "shared/BackgroundModel.h"
#include "shared/SequenceSet.h"

class BasePattern {
  friend class IUPACPattern;

 public:
  BasePattern(const size_t pattern_length, Strand s, const int k, const int max_k, SequenceSet* sequence_set,
              BackgroundModel* bg);
  ~BasePattern();

  void init(const size_t pattern_length);
  size_t* getFactors() { return factor; }
  size_t getPatternLength() { return pattern_length; }
  std::string toString(const size_t pattern_id);
  size_t getRevCompId(const size_t pattern_id);
  size_t getFastRevCompId(const size_t pattern_id);
  size_t getNumberPatterns() { return number_patterns; }
  int getBackgroundOrder() const { return k; }
  float* getExpectedCounts() { return host_expected(); }
  size_t* getPatternCounter() { return host_counts(); }
  float* getBackgroundProb(const int order) { return host_bgprob()[order]; }
  float* getBackgroundProb() { return host_bgprob()[k]; }
  size_t baseId2IUPACId(const size_t base_pattern);
  float getExpCountFraction(const size_t pattern, const size_t pseudo_expected_pattern_counts);
  float getLogPval(size_t pattern) { return host_logp()[pattern]; }
  float getOptimizationScore(const OPTIMIZATION_SCORE score_type, const size_t pattern,
                             const size_t pseudo_expected_pattern_counts);
  size_t getLtot() { return ltot; }
  Strand getStrand() const { return strand; }

  std::vector<size_t> select_base_patterns(const float zscore_threshold, const size_t count_threshold, bool single_stranded,
                                           bool filter_neighbors);
  std::vector<size_t> generate_double_stranded_em_optimization_patterns();
  void print_patterns(std::vector<size_t> patterns);

  inline size_t add_letter_to_the_right(size_t kmer, size_t position, int letter) { return kmer + letter * factor[position]; }

  // BaMM (big-endian) id of the (k+1)-mer that ends at position pattern_length-1 of a PEnG id
  inline size_t get_bg_id(const size_t pattern, const int pattern_length, const int k) {
    size_t y = 0;
    for (int i = pattern_length - k - 1; i < pattern_length; i++) y = y * 4 + (size_t)getNucleotideAtPos(pattern, i);
    return y;
  }
  inline size_t get_bg_id(const size_t pattern, const int pattern_length) {
    return get_bg_id(pattern, pattern_length, pattern_length - 1);
  }
  inline int getNucleotideAtPos(const size_t pattern, const size_t pos) { return (int)((pattern >> (2 * pos)) & 3); }
  inline int getFastNucleotideAtPos(const size_t pattern, const size_t pos) { return (int)((pattern >> (2 * pos)) & 3); }

  // device-resident tables for the kernels that run later (IUPAC aggregation, EM)
  const uint32_t* device_counts() const { return d_counts.get(); }
  const float* device_bgprob(int order) const { return d_bgprob.get() + (size_t)order * number_patterns; }
  const float* device_expected() const { return d_expected.get(); }

 private:
  float getMutualInformationScore(size_t pattern);

  // host mirrors (borrowed by the getters above, valid for the object's life), fetched on first use
  size_t* host_counts();
  const uint32_t* host_counts32();
  float** host_bgprob();
  float* host_logp();
  float* host_zscore();
  float* host_expected();
  float* fetch(const float* d_src, size_t n);
  void* mirror(const void* d_src, size_t bytes);
  void release_mirror(void* h, size_t bytes);

  size_t* factor;
  size_t pattern_length;
  size_t* pattern_counter = nullptr;
  uint32_t* counts32 = nullptr;
  float** pattern_bg_probabilities = nullptr;
  float* pattern_logp = nullptr;
  float* pattern_zscore = nullptr;
  float* expected_counts = nullptr;
  BackgroundModel* background_model;
  size_t number_patterns;
  int max_k;
  int k;
  Strand strand;
  int alphabet_size;
  size_t n_sequences;
  size_t ltot;

  pengk_host::DeviceBuffer<uint32_t> d_counts;
  pengk_host::DeviceBuffer<float> d_bgprob;
  pengk_host::DeviceBuffer<float> d_expected;
  pengk_host::DeviceBuffer<float> d_logp;
  pengk_host::DeviceBuffer<float> d_z;
};

class sort_indices {
 public:
  explicit sort_indices(float* parr) : mparr(parr) {}
  bool operator()(const size_t i, const size_t j) const { return mparr[i] > mparr[j]; }

 private:
  float* mparr;
};

#endif
