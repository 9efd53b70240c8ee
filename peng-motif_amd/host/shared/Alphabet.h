// Alphabet -- base <-> code tables of the host mirror (interface of the reference's
// src/shared/Alphabet.h: same static members, same codes: 0 = other, A,C,G,T = 1..4, either case).
#ifndef PENGK_HOST_ALPHABET_H_
#define PENGK_HOST_ALPHABET_H_

#include <stdint.h>

#include <cstring>
#include <iomanip>
#include <iostream>

class Alphabet {
 public:
  static void init(const char* alphabetType);  // only "STANDARD" is on the PEnG path
  static void destruct();
  static int getSize() { return size_; }
  static void setSize(int size) { size_ = size; }
  static char* getAlphabet() { return alphabet_; }
  static char* getComplementAlphabet() { return complement_; }
  static uint8_t getCode(char base) { return code_of_[(unsigned char)base]; }
  static char getBase(uint8_t code) { return base_of_[code]; }
  static uint8_t getComplementCode(uint8_t code) { return complement_of_[code]; }

 private:
  static int size_;
  static char alphabet_[8];
  static char complement_[8];
  static uint8_t code_of_[256];  // bytes >= 128 map to 0 (the reference indexes out of bounds there)
  static char base_of_[8];
  static uint8_t complement_of_[8];
};

#endif
