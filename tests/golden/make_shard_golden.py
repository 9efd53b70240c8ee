#!/usr/bin/env python3
"""What N ranks must produce TOGETHER, derived from the REAL reference shard by shard.

The compiled reference cannot run BASELINE configs[3] (100M x 200 bp: its positions are 32-bit) nor the
weak-scaled sets `bench.py --gpus N` holds at N > 1, but it can run every SHARD of them, and the count tables of shards
add exactly: the non-overlap rule never crosses a sequence boundary (/root/reference/src/base_pattern.cpp:382,438),
`ltot` and the 84 background counters are plain sums.  So for each job below this script runs `oracle/_ref/ref_dump` on
every shard, keeps the shard's table in a scratch directory (resumable: a shard is ~10 minutes and ~30 GB of host memory)
and writes to tests/golden/shard_prefix_checksums.json

  * one row per shard: sha256 of its (mirrored) uint32 count table, ltot, the 84 background counters -- all the
    reference's own numbers;
  * one row per prefix k = 1, 2, 4, 8: sha256 of the SUM of the first k tables, sum of ltot, sum of the counters (the
    reference's numbers, added), and -- derived from those sums by the oracle's sweep, which is pinned against the
    reference elsewhere and labelled "oracle on reference sums" here -- sha256 of V and of z.

tests/test_gpu_shards.py accumulates the shards on ONE GPU and must hit the k = 8 rows; the multi-rank CLI / bench runs
hit the row of their world size.  Build container only:  make -C oracle ref && python tests/golden/make_shard_golden.py
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as po  # noqa: E402

REF_DUMP = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
SCRATCH = os.environ.get("PENGK_SHARD_SCRATCH", "/tmp/pengk_shard_golden")
JOBS = [  # name, seed, n per shard, L, W, strand, shards
    # BASELINE configs[3]: 100M x 200 bp, W = 12, 8 shards of 12.5M sequences
    ("configs3", 1, 12_500_000, 200, 12, "BOTH", 8),
    # bench.py's weak scaling of BASELINE configs[2]: rank r holds sequences [r * 10M, (r + 1) * 10M)
    ("configs2_weak", 1, 10_000_000, 200, 10, "BOTH", 8),
]
PREFIXES = (1, 2, 4, 8)  # (k = 1: what one rank must produce -- shard 0 -- with the oracle's V and z beside it)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_shard(name, seed, n, L, W, strand, s):
    path = os.path.join(SCRATCH, "%s_shard%d.npz" % (name, s))
    if os.path.exists(path):
        return path
    t0 = time.time()
    seq0 = s * n
    codes, offs = po.synth(seed, seq0, n, L)
    tmp = tempfile.mkdtemp(prefix="shardref_", dir=SCRATCH)
    fa = os.path.join(tmp, "s.fa")
    lut = np.frombuffer(b"NACGT", dtype=np.uint8)
    rows = lut[codes].reshape(n, L)
    with open(fa, "wb") as f:
        for lo in range(0, n, 250_000):
            f.write(b"".join((">s%d\n" % (seq0 + i)).encode() + rows[i].tobytes() + b"\n" for i in range(lo, min(n, lo + 250_000))))
    del rows
    # the oracle's own count of the shard first (its buffers are released before the reference takes its ~30 GB)
    oc, olt = po.count(codes, offs, W, strand == "BOTH")
    oc = oc.astype(np.uint32)
    del codes, offs
    subprocess.check_call([REF_DUMP, fa, str(W), strand, tmp, "tables"], stderr=subprocess.DEVNULL)
    meta = dict(l.split() for l in open(os.path.join(tmp, "meta.txt")))
    counts = np.fromfile(os.path.join(tmp, "counts.u64"), np.uint64)
    assert int(counts.max()) < 2 ** 32
    counts = counts.astype(np.uint32)
    bgc = np.fromfile(os.path.join(tmp, "bgcounts.i32"), np.int32).astype(np.int64)
    ltot = int(meta["ltot"])
    assert olt == ltot and np.array_equal(oc, counts), "oracle != reference on shard %d of %s" % (s, name)
    np.savez(path + ".tmp.npz", counts=counts, bgcounts=bgc, ltot=np.int64(ltot))
    os.replace(path + ".tmp.npz", path)
    shutil.rmtree(tmp, ignore_errors=True)
    print("%s shard %d: ltot %d, %.0f s" % (name, s, ltot, time.time() - t0), flush=True)
    return path


def main():
    os.makedirs(SCRATCH, exist_ok=True)
    only = [a for a in sys.argv[1:] if not a.startswith("-")]
    out_path = os.path.join(HERE, "shard_prefix_checksums.json")
    out = json.load(open(out_path)) if os.path.exists(out_path) else []
    for name, seed, n, L, W, strand, shards in JOBS:
        if only and name not in only:
            continue
        out = [r for r in out if r["job"] != name]
        base = dict(job=name, seed=seed, n_per_shard=n, L=L, W=W, strand=strand)
        tot = np.zeros(4 ** W, np.uint64)
        bg = np.zeros(84, np.int64)
        lt = 0
        for s in range(shards):
            d = np.load(run_shard(name, seed, n, L, W, strand, s))
            out.append(dict(base, kind="shard", shard=s, seq0=s * n, ltot=int(d["ltot"]), sha_counts_u32=sha(d["counts"]),
                            bgcounts=d["bgcounts"].tolist(), source="compiled reference"))
            tot += d["counts"]
            bg += d["bgcounts"]
            lt += int(d["ltot"])
            k = s + 1
            if k in PREFIXES:
                assert int(tot.max()) < 2 ** 32
                # 64-bit counters: the reference's `int` counters and base count wrap beyond 2^31 bases
                # (src/shared/BackgroundModel.cpp:492-495) -- every row here but the 2.0e9-base one is beyond that; below
                # it the two agree bit for bit (asserted), and that row's V / z equal the reference's own
                # (tests/golden/synth_checksums.json)
                V = po.bg_V(bg, 2, wide=True)
                if int(bg[:4].sum()) < 2 ** 31:
                    assert V.tobytes() == po.bg_V(bg, 2).tobytes()
                bgp = po.bgprob(W, 2, V, strand == "BOTH")
                e, lp, z = po.stats(W, tot, bgp, lt)
                out.append(dict(base, kind="prefix", k=k, n_seq=k * n, ltot=lt, sha_counts_u32=sha(tot.astype(np.uint32)),
                                bgcounts=bg.tolist(), source="compiled reference, per-shard tables added",
                                sha_V=sha(V), sha_bgp2=sha(bgp), sha_expected=sha(e), sha_z=sha(z),
                                derived="V, bgp2, expected, z: the oracle's sweep on the reference's summed tables, background counters in "
                                        "64 bits (the reference's int counters wrap beyond 2^31 bases; identical below)"))
                print("%s prefix %d: ltot %d" % (name, k, lt), flush=True)
            json.dump(out, open(out_path, "w"), indent=1)
    json.dump(out, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
