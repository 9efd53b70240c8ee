// SequenceSet -- FASTA reader of the host mirror (public interface of src/shared/SequenceSet.h).
// Parsing quirks of the reference are kept (src/shared/SequenceSet.cpp:285-447): an unterminated last
// line is not seen, blank lines are skipped, a header without sequence is dropped with a warning,
// a space inside a sequence line or a sequence before any header ends the program with exit(1).
#ifndef PENGK_HOST_SEQUENCESET_H_
#define PENGK_HOST_SEQUENCESET_H_

#include <string>
#include <vector>

#include "Alphabet.h"
#include "Sequence.h"

class SequenceSet {
 public:
  SequenceSet(std::string sequenceFilepath, bool single_stranded = false, std::string intensityFilepath = "");
  ~SequenceSet();

  std::string getSequenceFilepath() { return path_; }
  std::vector<Sequence*> getSequences() { return sequences_; }
  const std::vector<Sequence*>& sequences() const { return sequences_; }  // no copy
  size_t getN() { return sequences_.size(); }
  unsigned int getMinL() { return minL_; }
  unsigned int getMaxL() { return maxL_; }
  float* getBaseFrequencies() { return base_freq_; }

 private:
  void readFASTA(bool single_stranded);
  std::string path_;
  std::vector<Sequence*> sequences_;
  unsigned int minL_, maxL_;
  float base_freq_[4];
};

#endif
