#!/usr/bin/env python3
"""End-to-end fixtures from the reference CLI (oracle/_ref/peng_motif_ref, built from /root/reference by
oracle/Makefile): MEME, JSON and stdout for BASELINE configs 1 and 2 and a few flag variants.
Run in the build container only:  make -C oracle ref && python tests/golden/make_cli_golden.py"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "peng_motif_ref")

def wrapper_flags(w):
    return ["-w", str(w), "-t", "10", "--count-threshold", "1", "--bg-model-order", "2", "--strand", "BOTH",
            "--optimization_score", "MUTUAL_INFO", "--enrich_pseudocount_factor", "0.005", "-a", "10000.0", "--em-threshold", "0.08",
            "--em-max-iterations", "100", "--max_merged_length", "14", "-b", "0.4", "--pseudo-counts", "10", "--threads", "1",
            "--minimum-processed-patterns", "25", "--max-optimized-patterns", "50"]


CASES = {
    # name: (fasta, extra args)
    "cli_mafk100_w8": ("MafK_100seqs.fasta", ["-w", "8"]),                          # BASELINE config 1
    "cli_mafk100_w6_plus_noem": ("MafK_100seqs.fasta", ["-w", "6", "--strand", "PLUS", "--no-em"]),
    "cli_mafk100_w8_logpval_nomerge": ("MafK_100seqs.fasta", ["-w", "8", "--optimization_score", "LOGPVAL", "--no-merging", "-t", "5"]),
    "cli_mafk_w10": ("MafK.fasta", ["-w", "10"]),                                    # BASELINE config 2
    "cli_mafk_w10_plus": ("MafK.fasta", ["-w", "10", "--strand", "PLUS"]),
    "cli_torture_w6": ("torture.fa", ["-w", "6", "-t", "3", "--count-threshold", "2"]),
    "cli_mafk100_w8_bg1_enrich": ("MafK_100seqs.fasta", ["-w", "8", "--bg-model-order", "1", "--optimization_score", "ENRICHMENT", "-t", "6"]),
    "cli_mafk_w10_bg0_nofilter": ("MafK.fasta", ["-w", "10", "--bg-model-order", "0", "--no-neighbor-filtering", "--max-optimized-patterns", "12",
                                                  "--use-default-pwm", "--em-max-iterations", "3", "-a", "500", "--pseudo-counts", "20"]),
    # W = 2 (round 4): with the default background order (min(W - 1, 2) = 1) a 2-mer is modelled exactly and nothing is
    # ever a seed; order 0 gives seeds, a hill-climb, PWMs and an EM over 16 patterns
    "cli_mafk_w2_bg0": ("MafK.fasta", ["-w", "2", "--bg-model-order", "0", "-t", "3", "--count-threshold", "1"]),
    "cli_mafk100_w2": ("MafK_100seqs.fasta", ["-w", "2"]),
    # The command line the reference's Python wrapper builds (scripts/shoot_peng.py:122-153), every flag in its order with
    # the WRAPPER's defaults -- which differ from the binary's (count threshold 1, 100 EM iterations, at least 25 processed
    # patterns: scripts/shoot_peng.py:48,66,88 against src/Global.cpp:39,31,55): -w 6 on the 100-sequence set is the run
    # of the reference's CI (.travis.yml:17-23), -w 10 on the full set the wrapper's own default.
    "cli_wrapper_mafk100_w6": ("MafK_100seqs.fasta", wrapper_flags(6)),
    "cli_wrapper_mafk_w10": ("MafK.fasta", wrapper_flags(10)),
}


def main():
    if not os.path.exists(REF):
        sys.exit("build the reference first: make -C oracle ref")
    out = os.path.join(HERE, "cli")
    os.makedirs(out, exist_ok=True)
    only = [a for a in sys.argv[1:] if not a.startswith("-")]
    for name, (fasta, extra) in CASES.items():
        if only and name not in only:
            continue
        meme = os.path.join(out, name + ".meme")
        js = os.path.join(out, name + ".json")
        cmd = [REF, os.path.join(HERE, fasta)] + extra + ["-o", meme, "-j", js]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        open(os.path.join(out, name + ".stdout"), "wb").write(r.stdout)
        open(os.path.join(out, name + ".args"), "w").write(" ".join([fasta] + extra) + "\n")
        print(name, "rc", r.returncode, "motifs", open(meme).read().count("MOTIF"))


if __name__ == "__main__":
    main()
