// SequenceSet -- FASTA reader of the host mirror (public interface of src/shared/SequenceSet.h).
// Parsing quirks of the reference are kept (src/shared/SequenceSet.cpp:285-447): an unterminated last
// line is not seen, blank lines are skipped, a header without sequence is dropped with a warning,
// a space inside a sequence line or a sequence before any header ends the program with exit(1).
//
// Ingest is built for 10^7-record inputs (83 % of the reference's wall time at BASELINE config 3,
// SURVEY.md 8f.1): the file is mmap'ed, headers are located and records translated by all host cores,
// and the codes of all records live in ONE contiguous buffer (codes() / offsets()) that the packer
// consumes without a copy.  Sequence objects are only materialised if getSequences() is called.
#ifndef PENGK_HOST_SEQUENCESET_H_
#define PENGK_HOST_SEQUENCESET_H_

#include <stdint.h>

#include <string>
#include <vector>

#include "Alphabet.h"
#include "Sequence.h"

class SequenceSet {
 public:
  SequenceSet(std::string sequenceFilepath, bool single_stranded = false, std::string intensityFilepath = "");
  ~SequenceSet();

  std::string getSequenceFilepath() { return path_; }
  // the warnings this set's constructor wrote to stderr (a caller that re-uses the set where the reference reads the
  // file again replays them)
  const std::string& diagnostics() const { return diagnostics_; }
  std::vector<Sequence*> getSequences();  // materialises views on first use
  size_t getN() { return offs_.size() - 1; }
  unsigned int getMinL() { return minL_; }
  unsigned int getMaxL() { return maxL_; }
  float* getBaseFrequencies() { return base_freq_; }

  // contiguous representation: codes of record i are codes()[offsets()[i] .. offsets()[i+1])
  const uint8_t* codes() const { return codes_; }
  const int64_t* offsets() const { return offs_.data(); }

 private:
  void readFASTA();
  std::string path_;
  std::string diagnostics_;
  bool single_stranded_;
  uint8_t* codes_ = nullptr;
  std::vector<int64_t> offs_;
  std::vector<std::string> headers_;
  std::vector<Sequence*> sequences_;
  bool materialised_ = false;
  unsigned int minL_, maxL_;
  float base_freq_[4];
};

#endif
