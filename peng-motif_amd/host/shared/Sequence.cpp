#include "Sequence.h"

#include <cstdlib>
#include <cstring>

Sequence::Sequence(uint8_t* sequence, int L, std::string header, std::vector<int>, bool singleStrand) {
  header_ = header;
  if (singleStrand) {
    L_ = L;
    codes_ = (uint8_t*)std::calloc(L_ ? L_ : 1, 1);
    std::memcpy(codes_, sequence, L_);
  } else {  // sequence, a separator 0, then the reverse complement (as the reference stores it)
    L_ = 2 * L + 1;
    codes_ = (uint8_t*)std::calloc(L_, 1);
    for (int i = 0; i < L; ++i) {
      codes_[i] = sequence[i];
      codes_[2 * L - i] = Alphabet::getComplementCode(sequence[i]);
    }
  }
}

Sequence* Sequence::view(uint8_t* codes, int L, std::string header) {
  Sequence* s = new Sequence();
  s->codes_ = codes;
  s->L_ = L;
  s->owns_ = false;
  s->header_ = std::move(header);
  return s;
}

Sequence::~Sequence() {
  if (owns_) std::free(codes_);
}

std::unique_ptr<uint8_t[]> Sequence::createReverseComplement() {
  std::unique_ptr<uint8_t[]> rc{new uint8_t[L_ ? L_ : 1]};
  for (int i = 0; i < L_; ++i) rc[i] = Alphabet::getComplementCode(codes_[L_ - 1 - i]);
  return rc;
}
