// utils.h -- scalar scores used by the hill-climb and the writers (same function names as the
// reference's src/utils.h).  Promotions follow SURVEY.md A.6: float variables, exp / log / log2 in double.
#ifndef PENGK_HOST_UTILS_H_
#define PENGK_HOST_UTILS_H_

#include <cmath>
#include <cstddef>

inline float calculate_entropy(float p) {
  return (float)(-(double)p * std::log((double)p) - (double)(1 - p) * std::log((double)(1 - p)));
}

inline float calculate_entropy_base2(float p) {
  return (float)(-(double)p * std::log2((double)p) - (double)(1 - p) * std::log2((double)(1 - p)));
}

inline float calculate_mutual_information_fast(float pattern_observed, float pattern_expected, unsigned int n_sequences,
                                               float prior) {
  const float p_obs = (float)(1 - std::exp((double)(-pattern_observed / (float)n_sequences)));
  const float p_exp = (float)(1 - std::exp((double)(-pattern_expected / (float)n_sequences)));
  const float q = prior;
  const float p = p_obs * q + p_exp * (1 - q);
  return -q * calculate_entropy(p_obs) - (1 - q) * calculate_entropy(p_exp) + calculate_entropy(p);
}

// -sum_q MI(q)/H(q) over q = .5, .1, .01; 0 when the pattern is not enriched (lower is better)
inline float mutual_information_score(float observed, float expected, unsigned int n_sequences) {
  if (observed < expected) return 0;
  float score = 0;
  const float priors[3] = {(float)0.5, (float)0.1, (float)0.01};
  for (float q : priors) score += calculate_mutual_information_fast(observed, expected, n_sequences, q) / calculate_entropy(q);
  return -score;
}

inline float calculate_pwm_info(float** pwm, unsigned length, unsigned n_states) {
  float total_info = 0;
  for (size_t pos = 0; pos < length; pos++)
    for (size_t state = 0; state < n_states; state++) {
      const float p = pwm[pos][state];
      if (p != 0) total_info = (float)((double)total_info + (double)p * std::log2((double)p));
    }
  return (float)((double)total_info + (double)length * std::log2((double)n_states));
}

#endif
