"""Debug helper: regenerate random case `seed` of test_cli_random_inputs_against_reference_binary, run both CLIs and
print a diff of their stdout traces.  usage (GPU box, repository root): python tests/tools/cli_random_case.py SEED [flags...]"""
import difflib
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CLI = os.path.join(ROOT, "peng-motif_amd", "host", "peng_motif")
REF = os.path.join(ROOT, "oracle", "_ref", "peng_motif_ref")
seed, flags = int(sys.argv[1]), sys.argv[2:]
rng = np.random.default_rng(500 + seed)
n, L = int(rng.integers(300, 1500)), int(rng.integers(60, 220))
motifs = ["".join(rng.choice(list("ACGT"), size=int(rng.integers(8, 13)))) for _ in range(3)]
lines = []
for i in range(n):
    s = rng.choice(list("ACGT"), size=int(L + rng.integers(-20, 21)))
    for m in motifs:
        if rng.random() < 0.35:
            mm = list(m)
            if rng.random() < 0.4:
                mm[int(rng.integers(0, len(mm)))] = str(rng.choice(list("ACGT")))
            at = int(rng.integers(0, len(s) - len(mm)))
            s[at:at + len(mm)] = mm
    if rng.random() < 0.05:
        s[int(rng.integers(0, len(s)))] = "N"
    t = "".join(s)
    if rng.random() < 0.1:
        t = t.lower()
    lines.append(">r%d\n%s\n" % (i, t))
tmp = tempfile.mkdtemp()
fa = os.path.join(tmp, "random.fa")
open(fa, "w").write("".join(lines))
out = {}
for tag, exe in (("ref", REF), ("here", CLI)):
    r = subprocess.run([exe, fa] + flags + ["-o", os.path.join(tmp, tag + ".meme")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    out[tag] = r.stdout.decode().split("\n")
    print(tag, "rc", r.returncode, "lines", len(out[tag]))
d = list(difflib.unified_diff(out["ref"], out["here"], "ref", "here", lineterm="", n=2))
print("\n".join(d[:120]))
print("motifs:", motifs)
