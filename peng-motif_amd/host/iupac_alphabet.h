// IUPACAlphabet -- the 11-letter degenerate alphabet A C G T S W R Y M K N (0..10): representatives
// (letter -> bases) and the mutation neighbourhood of the hill-climb (interface of the reference's
// src/iupac_alphabet.h; tables restated from SURVEY.md A.6).
#ifndef PENGK_HOST_IUPAC_ALPHABET_H_
#define PENGK_HOST_IUPAC_ALPHABET_H_

#include <cstddef>
#include <cstdint>
#include <vector>

const int IUPAC_ALPHABET_SIZE = 11;
enum class IUPAC_Alphabet { A = 0, C = 1, G = 2, T = 3, S = 4, W = 5, R = 6, Y = 7, M = 8, K = 9, N = 10 };

class IUPACAlphabet {
 public:
  static void init(const char* alphabet);
  static std::vector<int> get_similar_iupac_nucleotides(int c) { return similar_[c]; }
  static std::vector<int> get_representative_iupac_nucleotides(int c) { return representative_[c]; }
  static const std::vector<int>& similar(int c) { return similar_[c]; }
  static const std::vector<int>& representative(int c) { return representative_[c]; }
  static char getBase(int c);
  static int getCode(char c);
  static size_t getAlphabetSize() { return IUPAC_ALPHABET_SIZE; }

 private:
  static std::vector<int> similar_[IUPAC_ALPHABET_SIZE];
  static std::vector<int> representative_[IUPAC_ALPHABET_SIZE];
};

#endif
