// Sequence -- one FASTA record as byte codes (interface subset of src/shared/Sequence.h).
// The reference additionally precomputes a 9-mer int per base (Sequence.cpp:28-33) that only its
// BackgroundModel reads; here BackgroundModel derives the same counters directly from the codes.
// A Sequence either owns its codes (public constructor, as in the reference) or is a view into the
// contiguous code buffer of its SequenceSet (how the FASTA reader creates them).
#ifndef PENGK_HOST_SEQUENCE_H_
#define PENGK_HOST_SEQUENCE_H_

#include <stdint.h>

#include <memory>
#include <string>
#include <vector>

#include "Alphabet.h"

class Sequence {
 public:
  Sequence(uint8_t* sequence, int L, std::string header, std::vector<int> Y, bool singleStrand = false);
  static Sequence* view(uint8_t* codes, int L, std::string header);  // non-owning
  ~Sequence();
  uint8_t* getSequence() { return codes_; }
  int getL() { return L_; }
  std::string getHeader() { return header_; }
  std::unique_ptr<uint8_t[]> createReverseComplement();

 private:
  Sequence() = default;
  uint8_t* codes_ = nullptr;
  int L_ = 0;
  bool owns_ = true;
  std::string header_;
};

#endif
