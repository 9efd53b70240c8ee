#include "Global.h"

#include "device.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iostream>
#include <string>
#include <sys/time.h>

char* Global::alphabetType = (char*)"STANDARD";
char* Global::outputFilename = nullptr;
char* Global::jsonFilename = nullptr;
char* Global::inputSequenceFilename = nullptr;
char* Global::backgroundSequenceFilename = nullptr;
SequenceSet* Global::inputSequenceSet = nullptr;
SequenceSet* Global::backgroundSequenceSet = nullptr;
OPTIMIZATION_SCORE Global::optScoreType = OPTIMIZATION_SCORE::MutualInformation;
float Global::enrich_pseudocount_factor = 0.005f;
int Global::patternLength = 10;
Strand Global::strand = Strand::BOTH_STRANDS;
bool Global::useEm = true;
float Global::emSaturationFactor = 1E4;
float Global::emMinThreshold = 0.08f;
int Global::emMaxIterations = 10;
bool Global::useMerging = true;
int Global::pseudoCounts = 10;
bool Global::useAdvPWM = true;
float Global::zscoreThreshold = 10;
size_t Global::countThreshold = 3;
float Global::mergeBitfactorThreshold = 0.4f;
size_t Global::max_merged_length = 14;
bool Global::interpolateBG = true;
int Global::bgModelOrder = 2;
int Global::maxOptBgModelOrder = 2;
std::vector<float> Global::bgModelAlpha(3, 1.0f);
int Global::verbosity = 2;
int Global::nr_threads = 1;
int Global::device = 0;
bool Global::filter_neighbors = true;
unsigned Global::minimum_processed_motifs = 0;
int Global::maximum_optimized_patterns = 50;

void Global::init(int nargs, char* args[]) {
  readArguments(nargs, args);
  Alphabet::init(alphabetType);
  pengk_host::start_context();  // the device runtime starts while the FASTA files are read
  pengk_host::start_sharded_ingest();  // multi-GPU run: every rank reads its own byte range of the files
  // both strands are handled inside the count; sequences are always read single stranded
  // ... and every chunk of the input set is packed and sent to the device while the rest is still being read
  pengk_host::begin_streaming_pack(patternLength);
  inputSequenceSet = new SequenceSet(inputSequenceFilename, true);
  pengk_host::finish_streaming_pack(inputSequenceSet);
  // Without --background-sequences the reference reads the input file a second time (src/Global.cpp:66-75) and so
  // prints that file's warnings twice; the set is shared here, its warnings are replayed.
  if (backgroundSequenceFilename) {
    backgroundSequenceSet = new SequenceSet(backgroundSequenceFilename, true);
  } else {
    backgroundSequenceSet = inputSequenceSet;
    std::cerr << inputSequenceSet->diagnostics() << std::flush;
  }
}

void Global::destruct() {
  if (backgroundSequenceSet != inputSequenceSet) delete backgroundSequenceSet;
  delete inputSequenceSet;
  inputSequenceSet = backgroundSequenceSet = nullptr;
}

namespace {
// A line of the reference's logger (src/log.h:42-57, 119-130; only its argument parser uses it): "- HH:MM:SS.mmm LEVEL: "
// in front, and -- every call site ends its message with std::endl, the logger adds its own -- an empty line behind.
void log_line(const char* level, const std::string& msg) {
  char clock[11];
  time_t t;
  time(&t);
  tm r = {};
  strftime(clock, sizeof clock, "%X", localtime_r(&t, &r));
  struct timeval tv;
  gettimeofday(&tv, nullptr);
  fprintf(stderr, "- %s.%03ld %s: %s\n\n", clock, (long)tv.tv_usec / 1000, level, msg.c_str());
  fflush(stderr);
}
// value of an option that needs one; exits with 4 like the reference when it is missing
const char* need(int& i, int nargs, char* args[], void (*help)()) {
  const char* opt = args[i];
  if (++i >= nargs) {
    help();
    log_line("ERROR", std::string("No expression following ") + opt);
    exit(4);
  }
  return args[i];
}
}  // namespace

void Global::readArguments(int nargs, char* args[]) {
  if (nargs > 1 && !strcmp(args[1], "-h")) {
    printHelp();
    exit(0);
  }
  if (nargs > 1 && (!strcmp(args[1], "-version") || !strcmp(args[1], "--version"))) {
    std::cout << "peng_motif version " << VERSION_NUMBER << std::endl;
    exit(0);
  }
  if (nargs < 2) {
    fprintf(stderr, "Error: Arguments are missing! \n");
    printHelp();
    exit(-1);
  }
  inputSequenceFilename = args[1];
  for (int i = 2; i < nargs; i++) {
    const char* a = args[i];
    if (!strcmp(a, "-w")) {
      patternLength = std::stoi(need(i, nargs, args, printHelp));
      if (patternLength % 2 == 1) {
        log_line("ERROR", "Due to optimizations the pattern length has to be a multiple of 2");
        exit(4);
      }
    } else if (!strcmp(a, "--background-sequences")) {
      backgroundSequenceFilename = (char*)need(i, nargs, args, printHelp);
    } else if (!strcmp(a, "--optimization_score")) {
      const char* v = need(i, nargs, args, printHelp);
      if (!strcmp(v, "LOGPVAL")) optScoreType = OPTIMIZATION_SCORE::kLogPval;
      else if (!strcmp(v, "ENRICHMENT")) optScoreType = OPTIMIZATION_SCORE::kExpCounts;
      else if (!strcmp(v, "MUTUAL_INFO")) optScoreType = OPTIMIZATION_SCORE::MutualInformation;
      else {
        printHelp();
        log_line("ERROR", "Unknown expression following --optimization_score");
        exit(4);
      }
    } else if (!strcmp(a, "--enrich_pseudocount_factor")) {
      enrich_pseudocount_factor = std::stof(need(i, nargs, args, printHelp));
    } else if (!strcmp(a, "-v")) {
      verbosity = std::stoi(need(i, nargs, args, printHelp));
    } else if (!strcmp(a, "-o")) {
      outputFilename = (char*)need(i, nargs, args, printHelp);
    } else if (!strcmp(a, "-j")) {
      jsonFilename = (char*)need(i, nargs, args, printHelp);
    } else if (!strcmp(a, "-t")) {
      zscoreThreshold = std::stof(need(i, nargs, args, printHelp));
    } else if (!strcmp(a, "--count-threshold")) {
      countThreshold = std::stoi(need(i, nargs, args, printHelp));
    } else if (!strcmp(a, "-b")) {
      mergeBitfactorThreshold = std::stof(need(i, nargs, args, printHelp));
    } else if (!strcmp(a, "--use-default-pwm")) {
      useAdvPWM = false;
    } else if (!strcmp(a, "--pseudo-counts")) {
      pseudoCounts = std::stoi(need(i, nargs, args, printHelp));
    } else if (!strcmp(a, "--threads")) {
      nr_threads = std::stoi(need(i, nargs, args, printHelp));
    } else if (!strcmp(a, "--device")) {
      device = std::stoi(need(i, nargs, args, printHelp));
    } else if (!strcmp(a, "--no-em")) {
      useEm = false;
    } else if (!strcmp(a, "-a")) {
      emSaturationFactor = std::stof(need(i, nargs, args, printHelp));
    } else if (!strcmp(a, "--em-threshold")) {
      emMinThreshold = std::stof(need(i, nargs, args, printHelp));
    } else if (!strcmp(a, "--em-max-iterations")) {
      emMaxIterations = std::stoi(need(i, nargs, args, printHelp));
    } else if (!strcmp(a, "--no-merging")) {
      useMerging = false;
    } else if (!strcmp(a, "--max_merged_length")) {
      max_merged_length = std::stoi(need(i, nargs, args, printHelp));
    } else if (!strcmp(a, "--strand")) {
      const char* v = need(i, nargs, args, printHelp);
      if (!strcmp(v, "BOTH")) strand = Strand::BOTH_STRANDS;
      else if (!strcmp(v, "PLUS")) strand = Strand::PLUS_STRAND;
      else {
        printHelp();
        log_line("ERROR", "Unknown expression following --strand");
        exit(4);
      }
    } else if (!strcmp(a, "--bg-model-order")) {
      bgModelOrder = std::stoi(need(i, nargs, args, printHelp));
      if (bgModelOrder < 0 || bgModelOrder > 2) {
        log_line("ERROR", "background model orders above 2 are not supported");  // (not in the reference: its alpha vector has three entries, src/Global.cpp:49)
        exit(4);
      }
    } else if (!strcmp(a, "--no-neighbor-filtering")) {
      filter_neighbors = false;
    } else if (!strcmp(a, "--minimum-processed-patterns")) {
      minimum_processed_motifs = std::stoi(need(i, nargs, args, printHelp));
    } else if (!strcmp(a, "--max-optimized-patterns")) {
      maximum_optimized_patterns = std::stoi(need(i, nargs, args, printHelp));
    } else if (!strcmp(a, "--version")) {
      std::cout << "peng_motif " << VERSION_NUMBER << std::endl;  // (src/Global.cpp:299-301: without the word)
      exit(0);
    } else if (!strcmp(a, "-h")) {
      printHelp();
      exit(0);
    } else {
      if (pengk_host::rank() == 0) log_line("WARNING", std::string("Ignoring unknown option ") + a);
    }
  }
}

void Global::printHelp() {
  printf("\n");
  printf("SYNOPSIS:  peng_motif SEQFILE [OPTIONS]      (MI355X build: the k-mer count, z-score sweep,\n");
  printf("           IUPAC aggregation and EM run as HIP kernels through libpengk)\n\n");
  printf("  SEQFILE                        FASTA file with the input sequences\n");
  printf("  -o FILE                        write motifs in short MEME format\n");
  printf("  -j FILE                        write motifs as JSON\n");
  printf("  --background-sequences FILE    FASTA file for the background model (default: SEQFILE)\n");
  printf("  -w INT                         pattern length, even, 4..14 (default 10)\n");
  printf("  -t FLOAT                       z-score threshold for seed k-mers (default 10)\n");
  printf("  --count-threshold INT          minimum count of a seed k-mer (default 3)\n");
  printf("  --bg-model-order INT           background model order 0..2 (default 2)\n");
  printf("  --strand BOTH|PLUS             strands to search (default BOTH)\n");
  printf("  --optimization_score ENRICHMENT|LOGPVAL|MUTUAL_INFO   (default MUTUAL_INFO)\n");
  printf("  --enrich_pseudocount_factor F  pseudo count factor for ENRICHMENT (default 0.005)\n");
  printf("  --no-em                        skip the EM refinement\n");
  printf("  -a FLOAT                       EM saturation factor (default 1e4)\n");
  printf("  --em-threshold FLOAT           EM convergence threshold (default 0.08)\n");
  printf("  --em-max-iterations INT        (default 10)\n");
  printf("  --no-merging                   do not merge similar PWMs\n");
  printf("  -b FLOAT                       merge bit-factor threshold (default 0.4)\n");
  printf("  --max_merged_length INT        (default 14)\n");
  printf("  --use-default-pwm              simple PWM construction instead of the advanced one\n");
  printf("  --pseudo-counts INT            PWM pseudo counts (default 10)\n");
  printf("  --no-neighbor-filtering        keep Hamming-1 neighbours of selected seeds\n");
  printf("  --minimum-processed-patterns INT   (default 0)\n");
  printf("  --max-optimized-patterns INT       (default 50)\n");
  printf("  --threads INT                  accepted for compatibility\n");
  printf("  --device INT                   HIP device index (default 0)\n");
  printf("  -v INT                         verbosity\n");
  printf("  --version, -h\n\n");
}
