#!/bin/bash
# wall time of this repository's peng_motif on the golden inputs + a synthetic FASTA (GPU box)
R=$PWD
for args in "tests/golden/MafK_100seqs.fasta -w 8" "tests/golden/MafK.fasta -w 10" "tests/golden/MafK.fasta -w 10 --strand PLUS"; do
  echo "== $args"
  ( time PENGK_TIMING=1 peng-motif_amd/host/peng_motif $args -o /tmp/o.meme > /tmp/o.stdout ) 2>&1 | grep -E "timing|real"
done
python3 - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import oracle as po
n, L = int(os.environ.get("NSYN", "1000000")), 200
codes, offs = po.synth(1, 0, n, L)
lut = np.frombuffer(b"NACGT", dtype=np.uint8)
rows = lut[codes].reshape(n, L)
with open("/tmp/syn.fa", "wb") as f:
    for i in range(0, n, 100000):
        blk = rows[i:i + 100000]
        hdr = [(">s%d\n" % j).encode() for j in range(i, i + len(blk))]
        f.write(b"".join(h + r.tobytes() + b"\n" for h, r in zip(hdr, blk)))
print("wrote", os.path.getsize("/tmp/syn.fa"))
PY
echo "== synthetic ${NSYN:-1000000} x 200, W=10"
( time PENGK_TIMING=1 peng-motif_amd/host/peng_motif /tmp/syn.fa -w 10 -o /tmp/o.meme > /tmp/o.stdout ) 2>&1 | grep -E "timing|real"
grep -c MOTIF /tmp/o.meme
