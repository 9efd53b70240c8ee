#include "Alphabet.h"

#include <cctype>
#include <cstdlib>

int Alphabet::size_ = 0;
char Alphabet::alphabet_[8] = "";
char Alphabet::complement_[8] = "";
uint8_t Alphabet::code_of_[256];
char Alphabet::base_of_[8];
uint8_t Alphabet::complement_of_[8];

void Alphabet::init(const char* alphabetType) {
  if (std::strcmp(alphabetType, "STANDARD") != 0) {
    std::cerr << "Error: Correct alphabet type to STANDARD" << std::endl;
    exit(-1);
  }
  size_ = 4;
  std::strcpy(alphabet_, "ACGT");
  std::strcpy(complement_, "TGCA");
  std::memset(code_of_, 0, sizeof code_of_);
  std::memset(base_of_, 0, sizeof base_of_);
  std::memset(complement_of_, 0, sizeof complement_of_);
  for (int i = 0; i < size_; ++i) {
    code_of_[(unsigned char)alphabet_[i]] = (uint8_t)(i + 1);
    code_of_[(unsigned char)std::tolower(alphabet_[i])] = (uint8_t)(i + 1);
    base_of_[i + 1] = alphabet_[i];
  }
  for (int i = 0; i < size_; ++i) complement_of_[i + 1] = code_of_[(unsigned char)complement_[i]];
}

void Alphabet::destruct() { size_ = 0; }
