#include "SequenceSet.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <limits>

SequenceSet::SequenceSet(std::string sequenceFilepath, bool single_stranded, std::string intensityFilepath) {
  if (Alphabet::getSize() == 0) {
    std::cerr << "Error: Initialize Alphabet before constructing a SequenceSet" << std::endl;
    exit(-1);
  }
  path_ = sequenceFilepath;
  minL_ = std::numeric_limits<int>::max();
  maxL_ = 0;
  for (float& f : base_freq_) f = 0.f;
  readFASTA(single_stranded);
  if (!intensityFilepath.empty()) {
    std::cerr << "Error: SequenceSet::readIntensities() is not implemented so far." << std::endl;
    exit(1);
  }
}

SequenceSet::~SequenceSet() {
  for (Sequence* s : sequences_) delete s;
}

void SequenceSet::readFASTA(bool single_stranded) {
  FILE* f = std::fopen(path_.c_str(), "rb");
  if (!f) {
    std::cerr << "Error: Cannot open FASTA file: " << path_ << std::endl;
    exit(1);
  }
  std::string text;
  {
    char buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
  }
  std::fclose(f);

  unsigned long base_counts[4] = {0, 0, 0, 0};
  std::string header;
  std::vector<uint8_t> codes;
  bool open = false;
  auto finish = [&]() {
    if (!open) return;
    if (codes.empty()) {
      std::cerr << "Warning: Ignore FASTA entry without sequence: " << path_ << std::endl;
    } else {
      const unsigned L = (unsigned)codes.size();
      if (L > maxL_) maxL_ = L;
      if (L < minL_) minL_ = L;
      for (uint8_t c : codes)
        if (c) ++base_counts[c - 1];
      sequences_.push_back(new Sequence(codes.data(), (int)L, header, std::vector<int>(), single_stranded));
    }
    codes.clear();
    open = false;
  };
  size_t pos = 0;
  while (pos < text.size()) {
    const size_t nl = text.find('\n', pos);
    if (nl == std::string::npos) break;  // the reference's getline(...).good() never yields an unterminated last line
    const char* line = text.data() + pos;
    const size_t len = nl - pos;
    pos = nl + 1;
    if (len == 0) continue;
    if (line[0] == '>') {
      finish();
      open = true;
      header = len == 1 ? std::to_string(sequences_.size() + 1) : std::string(line + 1, len - 1);
    } else if (open) {
      if (std::memchr(line, ' ', len)) {
        std::cerr << "Error: FASTA sequence contains space character: " << path_ << std::endl;
        exit(1);
      }
      for (size_t i = 0; i < len; ++i) codes.push_back(Alphabet::getCode(line[i]));
    } else {
      std::cerr << "Error: Wrong FASTA format: " << path_ << std::endl;
      exit(1);
    }
  }
  finish();
  unsigned long total = base_counts[0] + base_counts[1] + base_counts[2] + base_counts[3];
  for (int i = 0; i < 4; ++i) base_freq_[i] = (float)base_counts[i] / (float)total;
}
