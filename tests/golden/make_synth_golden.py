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
    # BASELINE configs[3] (100M x 200 bp, W = 12, 8 shards): the head of shard 0 at the largest size the compiled
    # reference handles comfortably here (400 Mbp; its positions are 32-bit and its tables 8-byte)
    (1, 0, 2_000_000, 200, 12, "BOTH"),
    # ... and one WHOLE shard of it (shard 3 of 8: sequences [37.5M, 50M)): 2.65e9 positions, just inside the
    # reference's 32-bit position counters; about 30 GB of host memory and ten minutes
    (1, 37_500_000, 12_500_000, 200, 12, "BOTH"),
    # BASELINE configs[2] at FULL size (the bench's own workload: 10M x 200 bp, W = 10, both strands) and the PLUS table
    # configs[4] runs its EM stress on: 2.0e9 bases, inside the reference's `int` base counts, so V, the sweep and the
    # seed list are pinned as well; about 25 GB of host memory and six minutes each
    (1, 0, 10_000_000, 200, 10, "BOTH"),
    (1, 0, 10_000_000, 200, 10, "PLUS"),
]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    path = os.path.join(HERE, "synth_checksums.json")
    out = json.load(open(path)) if os.path.exists(path) and "--all" not in sys.argv else []
    have = {(r["seed"], r["seq0"], r["n_seq"], r["L"], r["W"], r["strand"]) for r in out}
    for seed, seq0, n, L, W, strand in CASES:
        if (seed, seq0, n, L, W, strand) in have:
            continue
        codes, offs = po.synth(seed, seq0, n, L)
        tmp = tempfile.mkdtemp(prefix="synthref_")
        fa = os.path.join(tmp, "s.fa")
        lut = np.frombuffer(b"NACGT", dtype=np.uint8)
        rows = lut[codes].reshape(n, L)
        with open(fa, "wb") as f:
            for lo in range(0, n, 250_000):  # in pieces: the whole-shard case is a 2.7 GB file
                f.write(b"".join((">s%d\n" % (seq0 + i)).encode() + rows[i].tobytes() + b"\n" for i in range(lo, min(n, lo + 250_000))))
        del rows
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
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
        print(W, strand, n, "ltot", rec["ltot"], "seeds", len(rec["seeds"]), flush=True)
    json.dump(out, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
