// ref_dump.cpp -- fixture generator that drives the REAL reference classes (built from the
// sources under /root/reference by oracle/Makefile into oracle/_ref/; never copied into this
// repository, never shipped to the GPU box).
//
// TEST INFRASTRUCTURE ONLY.  It exists to (a) pin oracle/peng_oracle.cpp and (b) produce the
// golden vectors committed under tests/golden/ (see tests/golden/make_golden.py).
//
// usage: ref_dump <fasta> <W> <BOTH|PLUS> <outdir> [tables]
// writes raw little-endian arrays + text tables into <outdir>.

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <limits>
#include <map>
#include <memory>
#include <numeric>
#include <set>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>

// the reference keeps its tables private; the fixture generator needs to read them.
#define private public
#include "base_pattern.h"
#include "iupac_alphabet.h"
#include "iupac_pattern.h"
#include "peng.h"
#undef private

template <class T>
static void dump(const std::string& path, const T* p, size_t n) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) {
    perror(path.c_str());
    exit(2);
  }
  fwrite(p, sizeof(T), n, f);
  fclose(f);
}

int main(int argc, char** argv) {
  if (argc < 5) {
    fprintf(stderr, "usage: %s fasta W BOTH|PLUS outdir\n", argv[0]);
    return 2;
  }
  const std::string fasta = argv[1];
  const int W = atoi(argv[2]);
  const Strand strand = strcmp(argv[3], "PLUS") == 0 ? Strand::PLUS_STRAND : Strand::BOTH_STRANDS;
  const std::string out = argv[4];
  const bool tables_only = argc > 5 && strcmp(argv[5], "tables") == 0;  // skip the IUPAC / EM stages (large inputs)

  Alphabet::init("STANDARD");
  SequenceSet* ss = new SequenceSet(fasta, true);
  std::vector<float> alpha{1.f, 1.f, 1.f};
  BackgroundModel* bg = new BackgroundModel(*ss, 2, alpha, true);
  IUPACAlphabet::init(Alphabet::getAlphabet());
  const int K = std::min(W - 1, 2);

  // ---- background model ---------------------------------------------------------------------
  {
    std::vector<int> n;
    std::vector<float> v;
    for (int k = 0; k <= 2; ++k)
      for (int y = 0; y < (1 << (2 * (k + 1))); ++y) {
        n.push_back(bg->n_[k][y]);
        v.push_back(bg->getV()[k][y]);
      }
    dump(out + "/bgcounts.i32", n.data(), n.size());
    dump(out + "/V.f32", v.data(), v.size());
  }

  // ---- sequence codes as the reference parsed them --------------------------------------------
  {
    std::vector<uint8_t> codes;
    std::vector<int64_t> offs(1, 0);
    for (Sequence* s : ss->getSequences()) {
      codes.insert(codes.end(), s->getSequence(), s->getSequence() + s->getL());
      offs.push_back((int64_t)codes.size());
    }
    dump(out + "/codes.u8", codes.data(), codes.size());
    dump(out + "/offs.i64", offs.data(), offs.size());
  }

  // ---- BasePattern tables -----------------------------------------------------------------------
  IUPACPattern::init(17, bg->getV()[0]);
  const auto t_bp0 = std::chrono::steady_clock::now();
  BasePattern* bp = new BasePattern(W, strand, K, K, ss, bg);
  const double t_bp = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_bp0).count();
  const size_t NP = bp->getNumberPatterns();
  {
    std::vector<uint64_t> c(NP);
    for (size_t i = 0; i < NP; ++i) c[i] = bp->pattern_counter[i];
    dump(out + "/counts.u64", c.data(), NP);
    for (int k = 0; k <= K; ++k) dump(out + "/bgp" + std::to_string(k) + ".f32", bp->pattern_bg_probabilities[k], NP);
    dump(out + "/expected.f32", bp->expected_counts, NP);
    dump(out + "/z.f32", bp->pattern_zscore, NP);
    dump(out + "/logp.f32", bp->pattern_logp, NP);
  }
  std::vector<size_t> seeds = bp->select_base_patterns(10.f, 3, strand == Strand::PLUS_STRAND, true);
  {
    std::vector<uint64_t> s(seeds.begin(), seeds.end());
    dump(out + "/seeds.u64", s.data(), s.size());
    std::vector<size_t> s2 = bp->select_base_patterns(3.f, 1, strand == Strand::PLUS_STRAND, false);
    std::vector<uint64_t> t(s2.begin(), s2.end());
    dump(out + "/seeds_nofilter_z3.u64", t.data(), t.size());
  }
  {
    FILE* f = fopen((out + "/meta.txt").c_str(), "w");
    fprintf(f, "W %d\nNP %zu\nltot %zu\nN %zu\nK %d\nstrand %s\nnseeds %zu\nbasepattern_seconds %.6f\n", W, NP, bp->getLtot(),
            ss->getN(), K, strand == Strand::PLUS_STRAND ? "PLUS" : "BOTH", seeds.size(), t_bp);
    fclose(f);
  }

  if (tables_only) return 0;
  // ---- IUPAC aggregation: every seed (first 12) and all of its single-letter mutants, plus a
  //      second generation from the first mutant of each position (more degenerate letters) -----
  {
    FILE* f = fopen((out + "/iupac.txt").c_str(), "w");
    std::set<size_t> done;
    auto emit = [&](size_t id) {
      if (!done.insert(id).second) return;
      IUPACPattern p(id, W);
      p.aggregate_attributes_from_basepatterns(bp);
      float mi = p.getOptimizationScore(OPTIMIZATION_SCORE::MutualInformation, 0, (unsigned)ss->getN());
      unsigned long cc = p.count_combined_occurences(bp, id);
      uint32_t b[5];
      float v[5] = {p.get_bg_p(), p.getExpectedCounts(), p.getZscore(), p.getLogPval(), mi};
      memcpy(b, v, sizeof b);
      fprintf(f, "%zu %zu %lu %08x %08x %08x %08x %08x\n", id, p.get_sites(), cc, b[0], b[1], b[2], b[3], b[4]);
    };
    size_t ns = std::min<size_t>(seeds.size(), 12);
    for (size_t s = 0; s < ns; ++s) {
      size_t id = bp->baseId2IUPACId(seeds[s]);
      emit(id);
      for (int p = 0; p < W; ++p) {
        int c = IUPACPattern::getNucleotideAtPos(id, p);
        size_t masked = id - c * IUPACPattern::iupac_factor[p];
        bool first = true;
        for (int r : IUPACAlphabet::get_similar_iupac_nucleotides(c)) {
          size_t m1 = masked + r * IUPACPattern::iupac_factor[p];
          emit(m1);
          if (first && s < 3) {
            first = false;
            for (int p2 = 0; p2 < W; ++p2) {
              int c2 = IUPACPattern::getNucleotideAtPos(m1, p2);
              size_t masked2 = m1 - c2 * IUPACPattern::iupac_factor[p2];
              for (int r2 : IUPACAlphabet::get_similar_iupac_nucleotides(c2)) emit(masked2 + r2 * IUPACPattern::iupac_factor[p2]);
            }
          }
        }
      }
    }
    // per-seed base-pattern MI score (src/base_pattern.cpp:184-200)
    for (size_t s = 0; s < ns; ++s) {
      float mi = bp->getOptimizationScore(OPTIMIZATION_SCORE::MutualInformation, seeds[s], 0);
      uint32_t b;
      memcpy(&b, &mi, 4);
      fprintf(f, "# basemi %zu %08x\n", seeds[s], b);
    }
    fclose(f);
  }
  delete bp;

  // ---- end-to-end up to the PWMs, without and with EM (no merging), via Peng::process ----------
  for (int use_em = 0; use_em <= 1; ++use_em) {
    Peng peng(strand, K, K, ss, bg);
    PengParameters prm;
    prm.max_pattern_length = W;
    prm.zscore_threshold = 10;
    prm.count_threshold = 3;
    prm.pseudo_counts = 10;
    prm.opt_score_type = OPTIMIZATION_SCORE::MutualInformation;
    prm.enrich_pseudocount_factor = 0.005f;
    prm.use_em = use_em;
    prm.em_saturation_factor = 1E4;
    prm.em_min_threshold = 0.08f;
    prm.em_max_iterations = 10;
    prm.use_merging = false;
    prm.bit_factor_merge_threshold = 0.4f;
    prm.adv_pwm = true;
    prm.minimum_processed_motifs = 0;
    prm.filter_neighbors = true;
    prm.max_optimized_patterns = 50;
    prm.max_merged_length = 14;
    std::vector<IUPACPattern*> res;
    std::streambuf* keep = std::cout.rdbuf();
    std::ofstream sink(out + (use_em ? "/stdout_em.txt" : "/stdout_noem.txt"));
    std::cout.rdbuf(sink.rdbuf());
    peng.process(prm, res);
    std::cout.rdbuf(keep);
    FILE* f = fopen((out + (use_em ? "/pwm_em.txt" : "/pwm_noem.txt")).c_str(), "w");
    for (IUPACPattern* p : res) {
      uint32_t b[3];
      float v[3] = {p->getLogPval(), p->get_bg_p(), p->getExpectedCounts()};
      memcpy(b, v, sizeof b);
      fprintf(f, "%zu %zu %08x %08x %08x", p->get_pattern(), p->get_sites(), b[0], b[1], b[2]);
      for (int q = 0; q < W; ++q)
        for (int a = 0; a < 4; ++a) {
          uint32_t u;
          memcpy(&u, &p->get_pwm()[q][a], 4);
          fprintf(f, " %08x", u);
        }
      fprintf(f, "\n");
    }
    fclose(f);
  }
  return 0;
}
