#!/usr/bin/env python3
"""Checksums of the REAL reference's tables on the synthetic generator of SURVEY.md 8d, at sizes beyond what
the per-element fixtures cover (W = 10 and W = 12, tens of Mbp).  Written to tests/golden/synth_checksums.json;
tests/test_gpu_parity.py regenerates the same sequences ON THE DEVICE and compares.
Build container only:  make -C oracle ref && python tests/golden/make_synth_golden.py"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as po  # noqa: E402

REF_DUMP = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
CASES = [  # seed, seq0, n_seq, L, W, strand
    (1, 0, 200_000, 200, 10, "BOTH"),
    (1, 5_000_000, 100_000, 200, 10, "PLUS"),
    (1, 0, 60_000, 200, 12, "BOTH"),
    (3, 777, 40_000, 150, 12, "PLUS"),
    (2, 10, 30_000, 90, 8, "BOTH"),
]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    out = []
    for seed, seq0, n, L, W, strand in CASES:
        codes, offs = po.synth(seed, seq0, n, L)
        tmp = tempfile.mkdtemp(prefix="synthref_")
        fa = os.path.join(tmp, "s.fa")
        lut = np.frombuffer(b"NACGT", dtype=np.uint8)
        rows = lut[codes].reshape(n, L)
        with open(fa, "wb") as f:
            f.write(b"".join((">s%d\n" % (seq0 + i)).encode() + rows[i].tobytes() + b"\n" for i in range(n)))
        subprocess.check_call([REF_DUMP, fa, str(W), strand, tmp, "tables"], stderr=subprocess.DEVNULL)
        ld = lambda f_, t: np.fromfile(os.path.join(tmp, f_), t)  # noqa: E731
        meta = dict(l.split() for l in open(os.path.join(tmp, "meta.txt")))
        counts = ld("counts.u64", np.uint64)
        rec = dict(seed=seed, seq0=seq0, n_seq=n, L=L, W=W, strand=strand, ltot=int(meta["ltot"]),
                   sha_counts_u32=sha(counts.astype(np.uint32)), sha_z=sha(ld("z.f32", np.float32)),
                   sha_expected=sha(ld("expected.f32", np.float32)), sha_bgp2=sha(ld("bgp2.f32", np.float32)),
                   sha_V=sha(ld("V.f32", np.float32)), bgcounts=ld("bgcounts.i32", np.int32).tolist(),
                   seeds=ld("seeds.u64", np.uint64).tolist()[:40])
        # the oracle agrees (pins it at these sizes too)
        oc, lt = po.count(codes, offs, W, strand == "BOTH")
        assert lt == rec["ltot"] and sha(oc.astype(np.uint32)) == rec["sha_counts_u32"], "oracle != reference"
        out.append(rec)
        print(W, strand, n, "ltot", rec["ltot"], "seeds", len(rec["seeds"]))
    json.dump(out, open(os.path.join(HERE, "synth_checksums.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
