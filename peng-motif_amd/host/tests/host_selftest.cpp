// host_selftest -- the encoding checks the reference's own gtest fixture pins
// (test/test_base_pattern.cpp:38-131: add_letter_to_the_right, reverse complement, PEnG->BaMM id
// conversion, IUPAC expansion, nucleotide extraction), run against this repository's classes.
// Needs a GPU (the BasePattern constructor runs the device pipeline).  usage: host_selftest FASTA
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <set>

#include "base_pattern.h"
#include "device.h"
#include "iupac_alphabet.h"
#include "iupac_pattern.h"
#include "peng.h"

static int failures = 0;
#define EXPECT(cond)                                                     \
  do {                                                                   \
    if (!(cond)) {                                                       \
      std::fprintf(stderr, "FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); \
      ++failures;                                                        \
    }                                                                    \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  Alphabet::init("STANDARD");
  IUPACAlphabet::init(Alphabet::getAlphabet());
  SequenceSet* ss = new SequenceSet(argv[1], true);
  BackgroundModel* bg = new BackgroundModel(*ss, 2, std::vector<float>{1.f, 1.f, 1.f}, true);
  IUPACPattern::init(17, bg->getV()[0]);
  BasePattern* bp = new BasePattern(4, Strand::BOTH_STRANDS, 2, 2, ss, bg);

  // check_kmer_extension_right: id("A") + C at 1 + G at 2 + T at 3 == id("ACGT")
  size_t kmer = 0;
  kmer = bp->add_letter_to_the_right(kmer, 1, 1);
  kmer = bp->add_letter_to_the_right(kmer, 2, 2);
  kmer = bp->add_letter_to_the_right(kmer, 3, 3);
  EXPECT(bp->toString(kmer) == "ACGT");
  EXPECT(kmer == 0 + 1 * 4 + 2 * 16 + 3 * 64);

  // check_reverse_complement / check_fast_reverse_complement
  for (size_t x = 0; x < bp->getNumberPatterns(); ++x) {
    const size_t r = bp->getRevCompId(x);
    EXPECT(bp->getFastRevCompId(x) == r);
    EXPECT(bp->getRevCompId(r) == x);
    std::string s = bp->toString(x), t = bp->toString(r);
    std::reverse(t.begin(), t.end());
    for (size_t i = 0; i < s.size(); ++i) EXPECT(Alphabet::getComplementCode(Alphabet::getCode(s[i])) == Alphabet::getCode(t[i]));
  }
  EXPECT(bp->toString(bp->getRevCompId(kmer)) == "ACGT");

  // check_bg_kmer_conversion: PEnG ACGT (little endian) -> BaMM id (big endian)
  EXPECT(bp->get_bg_id(kmer, 4) == 0 * 64 + 1 * 16 + 2 * 4 + 3);
  EXPECT(bp->get_bg_id(kmer, 4, 1) == 2 * 4 + 3);   // "GT"
  EXPECT(bp->get_bg_id(kmer, 3, 2) == 0 * 16 + 1 * 4 + 2);  // "ACG"

  // test_nucleotide_at
  for (size_t p = 0; p < 4; ++p) {
    EXPECT(bp->getNucleotideAtPos(kmer, p) == (int)p);
    EXPECT(bp->getFastNucleotideAtPos(kmer, p) == (int)p);
  }

  // iupac2base_patterns: stack expansion == level-wise expansion as sets, size = product of letter sets
  const char* pats[] = {"ANSW", "NNNN", "ACGT", "RYKM", "SWAN"};
  for (const char* ps : pats) {
    size_t id = 0;
    for (int i = 0; i < 4; ++i) id += IUPACAlphabet::getCode(ps[i]) * IUPACPattern::iupac_factor[i];
    EXPECT(IUPACPattern::toString(id, 4) == ps);
    IUPACPattern ip(id, 4);
    std::vector<size_t> a = ip.generate_base_patterns(bp, id), b;
    IUPACPattern::find_base_patterns(bp, id, 4, b);
    size_t expect = 1;
    for (int i = 0; i < 4; ++i) expect *= IUPACAlphabet::get_representative_iupac_nucleotides(IUPACAlphabet::getCode(ps[i])).size();
    EXPECT(a.size() == expect && b.size() == expect);
    EXPECT(std::set<size_t>(a.begin(), a.end()) == std::set<size_t>(b.begin(), b.end()));
    // device aggregation == host sum over the expansion (counts are integers: exact)
    unsigned long host_sum = 0;
    std::vector<size_t> ds = ip.basepatterns_from_iupac_double_stranded(bp, id);
    for (size_t i = 0; i < ds.size(); ++i)
      if (i == 0 || ds[i] != ds[i - 1]) host_sum += bp->getPatternCounter()[ds[i]];
    EXPECT(ip.count_combined_occurences(bp, id) == host_sum);
  }

  // BasePattern tables are self-consistent: mirrored counts, expected = bgprob * ltot
  for (size_t x = 0; x < bp->getNumberPatterns(); ++x) {
    EXPECT(bp->getPatternCounter()[x] == bp->getPatternCounter()[bp->getRevCompId(x)]);
    EXPECT(bp->getExpectedCounts()[x] == bp->getBackgroundProb()[x] * (float)bp->getLtot());
  }

  delete bp;
  delete bg;
  delete ss;
  pengk_host::shutdown();
  if (failures) {
    std::fprintf(stderr, "%d check(s) failed\n", failures);
    return 1;
  }
  std::puts("host_selftest ok");
  return 0;
}
