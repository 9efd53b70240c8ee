// Counting follows src/shared/Sequence.cpp:28-33 + src/shared/BackgroundModel.cpp:60-84: with an invalid
// base among the (up to) 9 positions [i-8, i] only an all-zero (k+1)-mer (invalid bases count as digit 0)
// is counted; otherwise every (k+1)-mer ending at i >= k is.  calculateV follows :490-530 in float32.
#include "BackgroundModel.h"

#include <cstdlib>
#include <iostream>

BackgroundModel::BackgroundModel(SequenceSet& sequenceSet, int order, std::vector<float> alpha, bool interpolate,
                                 std::vector<std::vector<int>>, std::vector<int>) {
  if (order < 0 || order > 2 || (int)alpha.size() < order + 1) {
    std::cerr << "Error: background model order " << order << " is not supported (0..2)" << std::endl;
    exit(-1);
  }
  const std::string path = sequenceSet.getSequenceFilepath();
  const size_t slash = path.find_last_of('/');
  name_ = slash == std::string::npos ? path : path.substr(slash + 1);
  const size_t dot = name_.find_last_of('.');
  if (dot != std::string::npos) name_ = name_.substr(0, dot);
  K_ = order;
  A_ = alpha;
  interpolate_ = interpolate;
  for (int k = 0; k < 3; ++k) {
    n_[k] = new long long[1 << (2 * (k + 1))]();
    v_[k] = new float[1 << (2 * (k + 1))]();
  }
  for (Sequence* s : sequenceSet.sequences()) {
    const uint8_t* seq = s->getSequence();
    const int L = s->getL();
    unsigned digits = 0, invalid = 0;  // rolling 9-base window
    for (int i = 0; i < L; ++i) {
      const unsigned c = seq[i];
      digits = ((digits << 2) | (c ? c - 1u : 0u)) & 0x3FFFFu;
      invalid = ((invalid << 1) | (c == 0)) & 0x1FFu;
      for (int k = 0; k <= K_ && k <= i; ++k) {
        const unsigned y = digits & ((1u << (2 * (k + 1))) - 1u);
        if (invalid == 0 || y == 0) ++n_[k][y];
      }
    }
  }
  calculateV();
}

BackgroundModel::~BackgroundModel() {
  for (int k = 0; k < 3; ++k) {
    delete[] n_[k];
    delete[] v_[k];
  }
}

void BackgroundModel::calculateV() {
  int base_counts = 0;  // `int` like the reference
  for (int y = 0; y < 4; ++y) base_counts += (int)n_[0][y];
  for (int y = 0; y < 4; ++y)
    v_[0][y] = ((float)(int)n_[0][y] + A_[0] * 0.25f) / ((float)base_counts + A_[0]);
  for (int k = 1; k <= K_; ++k) {
    const int ny = 1 << (2 * (k + 1)), yk = 1 << (2 * k);
    for (int y = 0; y < ny; ++y) {
      const float prior = interpolate_ ? v_[k - 1][y % yk] : 0.25f;
      v_[k][y] = ((float)(int)n_[k][y] + A_[k] * prior) / ((float)(int)n_[k - 1][y / 4] + A_[k]);
    }
    for (int g = 0; g < ny; g += 4) {
      float factor = 0.0f;
      for (int a = 0; a < 4; ++a) factor += v_[k][g + a];
      for (int a = 0; a < 4; ++a) v_[k][g + a] /= factor;
    }
  }
}
