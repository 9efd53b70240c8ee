// Global -- command-line state of the peng_motif mirror (same static fields, flags, defaults and exit
// codes as the reference's src/Global.h / src/Global.cpp:77-314).
#ifndef PENGK_HOST_GLOBAL_H_
#define PENGK_HOST_GLOBAL_H_

#include <string>
#include <vector>

#include "shared/SequenceSet.h"

enum class Strand { PLUS_STRAND, BOTH_STRANDS };

enum class OPTIMIZATION_SCORE { kLogPval = 0, kExpCounts = 1, MutualInformation };

const std::string VERSION_NUMBER("1.0.0");

class Global {
 public:
  static char* alphabetType;
  static char* outputFilename;               // -o  short MEME
  static char* jsonFilename;                 // -j  JSON
  static char* inputSequenceFilename;        // positional
  static char* backgroundSequenceFilename;   // --background-sequences
  static SequenceSet* inputSequenceSet;
  static SequenceSet* backgroundSequenceSet;

  static OPTIMIZATION_SCORE optScoreType;    // --optimization_score
  static float enrich_pseudocount_factor;    // --enrich_pseudocount_factor

  static int patternLength;                  // -w (even)
  static float zscoreThreshold;              // -t
  static size_t countThreshold;              // --count-threshold
  static Strand strand;                      // --strand

  static bool useEm;                         // --no-em
  static float emSaturationFactor;           // -a
  static float emMinThreshold;               // --em-threshold
  static int emMaxIterations;                // --em-max-iterations

  static bool useMerging;                    // --no-merging
  static float mergeBitfactorThreshold;      // -b
  static size_t max_merged_length;           // --max_merged_length

  static bool useAdvPWM;                     // --use-default-pwm
  static int pseudoCounts;                   // --pseudo-counts

  static int bgModelOrder;                   // --bg-model-order
  static int maxOptBgModelOrder;
  static bool interpolateBG;
  static std::vector<float> bgModelAlpha;

  static int nr_threads;                     // --threads (accepted; the device path ignores it)
  static int verbosity;                      // -v
  static int device;                         // --device (new: HIP device index, default 0)

  static bool filter_neighbors;              // --no-neighbor-filtering
  static unsigned minimum_processed_motifs;  // --minimum-processed-patterns
  static int maximum_optimized_patterns;     // --max-optimized-patterns

  static void init(int nargs, char* args[]);
  static void destruct();

 private:
  static void readArguments(int nargs, char* args[]);
  static void printHelp();
};

#endif
