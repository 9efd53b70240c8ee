#include "iupac_alphabet.h"

#include <cstring>

std::vector<int> IUPACAlphabet::similar_[IUPAC_ALPHABET_SIZE];
std::vector<int> IUPACAlphabet::representative_[IUPAC_ALPHABET_SIZE];

namespace {
const char kLetters[] = "ACGTSWRYMKN";
enum { A, C, G, T, S, W, R, Y, M, K, N };
}  // namespace

void IUPACAlphabet::init(const char*) {
  // one step up or sideways in the degeneracy lattice, always ending with N
  similar_[A] = {W, R, M, N};
  similar_[C] = {S, Y, M, N};
  similar_[G] = {S, R, K, N};
  similar_[T] = {W, Y, K, N};
  similar_[S] = {C, G, R, Y, M, K, N};
  similar_[W] = {A, T, R, Y, M, K, N};
  similar_[R] = {A, G, S, W, M, K, N};
  similar_[Y] = {C, T, S, W, M, K, N};
  similar_[M] = {A, C, S, W, R, Y, N};
  similar_[K] = {G, T, S, W, R, Y, N};
  similar_[N] = {A, C, G, T, S, W, R, Y, M, K};
  representative_[A] = {A};
  representative_[C] = {C};
  representative_[G] = {G};
  representative_[T] = {T};
  representative_[S] = {C, G};
  representative_[W] = {A, T};
  representative_[R] = {A, G};
  representative_[Y] = {C, T};
  representative_[M] = {A, C};
  representative_[K] = {G, T};
  representative_[N] = {A, C, G, T};
}

char IUPACAlphabet::getBase(int c) { return (c >= 0 && c < IUPAC_ALPHABET_SIZE) ? kLetters[c] : 0; }

int IUPACAlphabet::getCode(char c) {
  const char* p = std::strchr(kLetters, c);
  return (p && c) ? (int)(p - kLetters) : 0;
}
