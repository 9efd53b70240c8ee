"""Differential sweep (GPU box, needs oracle/_ref/peng_motif_ref): random FASTA inputs and random flag combinations through
the compiled reference and through this repository's peng_motif; MEME, JSON and stdout must be identical byte for byte.
usage: [SCALE=100] python tests/tools/cli_differential.py FIRST_SEED N_CASES   (SCALE multiplies the number of sequences)"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CLI = os.path.join(ROOT, "peng-motif_amd", "host", "peng_motif")
REF = os.path.join(ROOT, "oracle", "_ref", "peng_motif_ref")


def make_case(seed, tmp):
    rng = np.random.default_rng(9000 + seed)
    W = int(rng.choice([int(x) for x in os.environ["PENGK_DIFF_WS"].split(",")] if os.environ.get("PENGK_DIFF_WS") else [4, 6, 8, 8, 10, 10, 12]))
    n, L = int(rng.integers(100, 1200)) * int(os.environ.get("SCALE", "1")), int(rng.integers(max(W + 2, 30), 260))
    motifs = ["".join(rng.choice(list("ACGT"), size=int(rng.integers(6, 14)))) for _ in range(int(rng.integers(1, 5)))]
    p_n = float(rng.choice([0.0, 0.02, 0.2]))

    def fasta(path, n_seq, with_motifs):
        lines = []
        for i in range(n_seq):
            s = rng.choice(list("ACGT"), size=max(3, int(L + rng.integers(-25, 26))))
            if with_motifs:
                for m in motifs:
                    if rng.random() < 0.4 and len(s) > len(m) + 1:
                        mm = list(m)
                        if rng.random() < 0.4:
                            mm[int(rng.integers(0, len(mm)))] = str(rng.choice(list("ACGT")))
                        at = int(rng.integers(0, len(s) - len(mm)))
                        s[at:at + len(mm)] = mm
            if rng.random() < p_n:
                s[int(rng.integers(0, len(s)))] = "N"
            t = "".join(s)
            if rng.random() < 0.1:
                t = t.lower()
            if rng.random() < 0.1:  # multi-line record
                cut = len(t) // 2
                t = t[:cut] + "\n" + t[cut:]
            lines.append(">r%d some text\n%s\n" % (i, t))
        open(path, "w").write("".join(lines))

    fa = os.path.join(tmp, "in.fa")
    fasta(fa, n, True)
    flags = ["-w", str(W)]
    if rng.random() < 0.4:
        flags += ["--strand", "PLUS"]
    if rng.random() < 0.3:
        flags += ["--bg-model-order", str(int(rng.integers(0, 3)))]
    if rng.random() < 0.3:
        flags += ["-t", str(float(rng.choice([3, 5, 8, 15])))]
    if rng.random() < 0.3:
        flags += ["--count-threshold", str(int(rng.integers(1, 6)))]
    if rng.random() < 0.3:
        flags += ["--optimization_score", str(rng.choice(["ENRICHMENT", "LOGPVAL", "MUTUAL_INFO"]))]
    if rng.random() < 0.2:
        flags += ["--enrich_pseudocount_factor", str(float(rng.choice([0.001, 0.01, 0.05])))]
    if rng.random() < 0.2:
        flags += ["--no-em"]
    if rng.random() < 0.2:
        flags += ["--no-merging"]
    if rng.random() < 0.2:
        flags += ["-b", str(float(rng.choice([0.2, 0.5, 0.7])))]
    if rng.random() < 0.2:
        flags += ["-a", str(float(rng.choice([100, 1e3, 1e5])))]
    if rng.random() < 0.2:
        flags += ["--em-threshold", str(float(rng.choice([0.0, 0.01, 0.5])))]
    if rng.random() < 0.2:
        flags += ["--em-max-iterations", str(int(rng.integers(1, 15)))]
    if rng.random() < 0.2:
        flags += ["--use-default-pwm"]
    if rng.random() < 0.2:
        flags += ["--pseudo-counts", str(int(rng.integers(1, 30)))]
    if rng.random() < 0.2:
        flags += ["--no-neighbor-filtering"]
    if rng.random() < 0.2:
        flags += ["--max-optimized-patterns", str(int(rng.integers(1, 30)))]
    if rng.random() < 0.2:
        flags += ["--minimum-processed-patterns", str(int(rng.integers(0, 10)))]
    if rng.random() < 0.2:
        flags += ["--max_merged_length", str(int(rng.integers(W, 20)))]
    if rng.random() < 0.2:
        bgf = os.path.join(tmp, "bg.fa")
        fasta(bgf, int(rng.integers(50, 500)), False)
        flags += ["--background-sequences", bgf]
    return fa, flags


def run(exe, fa, flags, tmp, tag):
    meme, js = os.path.join(tmp, tag + ".meme"), os.path.join(tmp, tag + ".json")
    r = subprocess.run([exe, fa] + flags + ["-o", meme, "-j", js], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    rd = lambda p: open(p).read() if os.path.exists(p) else None
    return r.returncode, r.stdout.decode(), r.stderr.decode(), rd(meme), rd(js)


first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, first + count):
    tmp = tempfile.mkdtemp()
    fa, flags = make_case(seed, tmp)
    a = run(REF, fa, flags, tmp, "ref")
    b = run(CLI, fa, flags, tmp, "here")
    same = [a[0] == b[0], a[1] == b[1], a[3] == b[3], a[4] == b[4]]
    status = "ok" if all(same) else "DIFF rc/stdout/meme/json=%s" % same
    if not all(same):
        bad += 1
    print(seed, status, "rc", a[0], " ".join(flags).replace(tmp, "."), "motifs", (a[3] or "").count("MOTIF"), flush=True)
print("differing cases:", bad)
sys.exit(1 if bad else 0)
