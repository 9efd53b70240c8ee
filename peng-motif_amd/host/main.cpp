// peng_motif -- command-line entry of the MI355X mirror: same flags, stdout trace, MEME / JSON output
// and exit codes as the reference's src/main.cpp; the hot loops run on the GPU through libpengk.
#include <algorithm>
#include <chrono>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "Global.h"
#include "device.h"
#include "iupac_pattern.h"
#include "peng.h"

namespace {
// PENGK_TIMING=1: wall-clock phase report on stderr (stdout stays the reference's trace)
struct PhaseClock {
  using clk = std::chrono::steady_clock;
  bool on = std::getenv("PENGK_TIMING") != nullptr;
  clk::time_point t0 = clk::now(), last = t0;
  void lap(const char* what) {
    if (!on) return;
    const auto now = clk::now();
    std::cerr << "[timing] " << what << ": " << std::chrono::duration<double>(now - last).count() << " s" << std::endl;
    last = now;
  }
  void total() {
    if (!on) return;
    std::cerr << "[timing] total: " << std::chrono::duration<double>(clk::now() - t0).count() << " s" << std::endl;
    // peak resident set of this process (a rank of a sharded run holds its part of the input only)
    if (FILE* f = std::fopen("/proc/self/status", "r")) {
      char line[256];
      while (std::fgets(line, sizeof line, f))
        if (std::strncmp(line, "VmHWM:", 6) == 0)
          std::cerr << "[timing] peak resident memory: " << std::atof(line + 6) / 1024.0 << " MiB" << std::endl;
        else if (std::strncmp(line, "VmRSS:", 6) == 0)  // (what is left for the kernel to take back when the process ends)
          std::cerr << "[timing] resident memory at the end: " << std::atof(line + 6) / 1024.0 << " MiB" << std::endl;
      std::fclose(f);
    }
    // how much of it sits on transparent huge pages: what the kernel has to take back page by page at exit is the rest
    if (FILE* f = std::fopen("/proc/self/smaps_rollup", "r")) {
      char line[256];
      while (std::fgets(line, sizeof line, f))
        if (std::strncmp(line, "AnonHugePages:", 14) == 0)
          std::cerr << "[timing] of it on transparent huge pages: " << std::atof(line + 14) / 1024.0 << " MiB" << std::endl;
      std::fclose(f);
    }
  }
};
}  // namespace

int main(int nargs, char** args) {
  PhaseClock clock;
  // multi-GPU run (one process per GPU under a launcher): every rank reads its byte range of the input, counts its
  // shard and follows the same deterministic control flow on the all-reduced tables; rank 0 alone reports and writes
  if (pengk_host::rank() != 0 && !std::freopen("/dev/null", "w", stdout)) return 1;
  Global::init(nargs, args);
  clock.lap("read FASTA (+ pack + upload, chunk by chunk)");
  // (the device context is being created on a helper thread since Global::init; nothing waits for it before the packed
  // sequences are ready for upload -- BasePattern's first device call -- and a machine without a gfx950 device fails
  // there, loudly: there is no CPU path)

  const int bg_model_order = std::max(Global::bgModelOrder, Global::maxOptBgModelOrder);
  // the input set doubles as the background set unless --background-sequences names another file: its (k+1)-mer
  // counters were taken by the packer while the file was read
  const pengk_host::PackedInput* packed = Global::backgroundSequenceSet == Global::inputSequenceSet
                                              ? pengk_host::packed_input(Global::inputSequenceSet, Global::patternLength)
                                              : nullptr;
  BackgroundModel* bgModel =
      packed ? new BackgroundModel(*Global::backgroundSequenceSet, bg_model_order, Global::bgModelAlpha, Global::interpolateBG,
                                   packed->bg_counts)
             : new BackgroundModel(*Global::backgroundSequenceSet, bg_model_order, Global::bgModelAlpha, Global::interpolateBG);

  clock.lap("background model");
  Peng peng(Global::strand, Global::bgModelOrder, Global::maxOptBgModelOrder, Global::inputSequenceSet, bgModel);

  PengParameters params;
  params.max_pattern_length = Global::patternLength;
  params.zscore_threshold = Global::zscoreThreshold;
  params.count_threshold = Global::countThreshold;
  params.pseudo_counts = Global::pseudoCounts;
  params.opt_score_type = Global::optScoreType;
  params.use_em = Global::useEm;
  params.em_saturation_factor = Global::emSaturationFactor;
  params.em_min_threshold = Global::emMinThreshold;
  params.em_max_iterations = Global::emMaxIterations;
  params.use_merging = Global::useMerging;
  params.bit_factor_merge_threshold = Global::mergeBitfactorThreshold;
  params.max_merged_length = Global::max_merged_length;
  params.adv_pwm = Global::useAdvPWM;
  params.enrich_pseudocount_factor = Global::enrich_pseudocount_factor;
  params.minimum_processed_motifs = Global::minimum_processed_motifs;
  params.filter_neighbors = Global::filter_neighbors;
  params.max_optimized_patterns = Global::maximum_optimized_patterns;

  std::vector<IUPACPattern*> result;
  peng.process(params, result);
  clock.lap("process (count, sweep, hill-climb, PWMs, EM, merging)");
  peng.filter_redundancy(Global::mergeBitfactorThreshold, result);
  if (pengk_host::rank() == 0) {
    if (Global::outputFilename) peng.printShortMeme(result, Global::outputFilename, bgModel);
    if (Global::jsonFilename) peng.printJson(result, Global::jsonFilename, VERSION_NUMBER, bgModel);
  }

  for (IUPACPattern* p : result) delete p;
  delete bgModel;
  clock.lap("output");
  // Everything this process owns -- gigabytes of sequence codes, the pinned table mirrors, the device buffers and the
  // HIP context -- dies with it; walking it all to hand it back piece by piece cost a quarter of the whole run on a
  // 2 GB input (0.25 of 1.1 s).  The outputs are closed and flushed, the device is idle: leave.
  // (PENGK_FULL_TEARDOWN=1 keeps the orderly release, e.g. under a leak checker.)
  pengk_host::check(pengk_synchronize(pengk_host::context()), "pengk_synchronize");
  pengk_host::finish_ranks();
  const char* full_teardown = std::getenv("PENGK_FULL_TEARDOWN");
  if (full_teardown && std::atoi(full_teardown)) {
    Global::destruct();
    pengk_host::shutdown();
    clock.lap("cleanup");
    clock.total();
    return 0;
  }
  clock.total();
  std::cout.flush();
  std::cerr.flush();
  fflush(nullptr);
  _exit(0);
}
