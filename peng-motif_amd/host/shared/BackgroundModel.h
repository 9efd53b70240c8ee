// BackgroundModel -- homogeneous interpolated Markov model of order K <= 2 (public interface subset of
// src/shared/BackgroundModel.h that PEnG uses: the counting constructor, getV, getOrder, getName).
#ifndef PENGK_HOST_BACKGROUNDMODEL_H_
#define PENGK_HOST_BACKGROUNDMODEL_H_

#include <string>
#include <vector>

#include "SequenceSet.h"

class BackgroundModel {
 public:
  BackgroundModel(SequenceSet& sequenceSet, int order, std::vector<float> alpha, bool interpolate = true,
                  std::vector<std::vector<int>> foldIndices = std::vector<std::vector<int>>(),
                  std::vector<int> folds = std::vector<int>());
  // The same model from (k+1)-mer counters somebody else has taken over this rank's records of `sequenceSet` -- the
  // packer counts them on its way (pengk_packed.bg_counts, exactly the counters of src/shared/BackgroundModel.cpp:60-84):
  // 84 values, orders 0..2 back to back, BaMM ids.  Summed over the ranks like the counting constructor's.
  BackgroundModel(SequenceSet& sequenceSet, int order, std::vector<float> alpha, bool interpolate, const long long* counts84);
  ~BackgroundModel();

  std::string getName() { return name_; }
  int getOrder() { return K_; }
  float** getV() { return v_; }
  const long long* getCounts(int k) const { return n_[k]; }  // (k+1)-mer counts, BaMM (big-endian) ids

 private:
  void init(SequenceSet& sequenceSet, int order, std::vector<float> alpha, bool interpolate);
  void calculateV();
  std::string name_;
  int K_;
  std::vector<float> A_;
  bool interpolate_;
  long long* n_[3];
  float* v_[3];
};

#endif
