"""What N ranks must produce TOGETHER, checked on ONE GPU against the compiled reference (SURVEY.md 8e:
`8-GPU == 1-GPU == sum of the per-shard reference counts`).

tests/golden/shard_prefix_checksums.json (tests/golden/make_shard_golden.py) holds, for BASELINE configs[3] (100M x 200 bp,
W = 12, eight shards of 12.5M sequences) and for the weak-scaled sets `bench.py --gpus N` holds (rank r = sequences
[r * 10M, (r + 1) * 10M), W = 10), the reference's own table of EVERY shard and the sums of the first 2, 4 and 8 of them --
count tables add exactly because the non-overlap rule never crosses a sequence boundary
(/root/reference/src/base_pattern.cpp:382,438).  Here the shards are generated and counted one after the other through
pengk_count_bg, each compared with its row, accumulated, and the accumulated {counts, ltot, background counters} go
through mirror -> V -> sweep on the device and must hit the prefix rows (V and z: the oracle's sweep on the reference's
sums).  The multi-rank CLI runs (tests/test_gpu_multirank.py) and bench.py (`checks_ok`) hit the same rows over the
exchange step."""
import hashlib
import json
import os

import numpy as np
import pytest

import peng_motif_amd as pk

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROWS = json.load(open(os.path.join(ROOT, "tests", "golden", "shard_prefix_checksums.json")))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def rows_of(job, kind):
    return [r for r in ROWS if r["job"] == job and r["kind"] == kind]


@pytest.fixture(scope="module")
def ctx():
    c = pk.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("job", ["configs3", "configs2_weak"])
def test_shards_counted_one_after_the_other_add_up_to_the_reference_sums(ctx, job):
    shards = sorted(rows_of(job, "shard"), key=lambda r: r["shard"])
    prefixes = {r["k"]: r for r in rows_of(job, "prefix")}
    assert len(shards) == 8 and sorted(prefixes) == [1, 2, 4, 8]
    W, L, n, both = shards[0]["W"], shards[0]["L"], shards[0]["n_per_shard"], shards[0]["strand"] == "BOTH"
    NP = 4 ** W
    tot = np.zeros(NP, np.uint64)
    bg_tot = np.zeros(84, np.uint64)
    lt_tot = 0
    words = items = None
    for row in shards:
        words, items, _, _ = ctx.synth(row["seed"], row["seq0"], n, L, W, 0, words, items)
        counts, ltot, bg = ctx.count_bg(both)
        c = counts.to_host()
        assert int(ltot.to_host()[0]) == row["ltot"]
        assert bg.to_host().tolist() == row["bgcounts"]
        tot += c
        bg_tot += bg.to_host()
        lt_tot += row["ltot"]
        if both:
            ctx.mirror(W, counts)
        assert sha(counts.to_host()) == row["sha_counts_u32"], "shard %d of %s" % (row["shard"], job)
        k = row["shard"] + 1
        if k in prefixes:
            want = prefixes[k]
            assert lt_tot == want["ltot"] and bg_tot.tolist() == want["bgcounts"]
            assert int(tot.max()) < 2 ** 32
            d_counts = ctx.to_device(tot.astype(np.uint32))
            if both:
                ctx.mirror(W, d_counts)
            assert sha(d_counts.to_host()) == want["sha_counts_u32"], "first %d shards of %s" % (k, job)
            V = ctx.bg_model(ctx.to_device(bg_tot), 2)
            d_ltot = ctx.to_device(np.array([lt_tot], np.uint64))
            bgprob, expected, logp, z = ctx.pattern_stats(W, both, 2, 2, V, d_ltot, d_counts)
            assert sha(V.to_host()) == want["sha_V"]
            assert sha(bgprob.to_host()[2]) == want["sha_bgp2"]
            assert sha(expected.to_host()) == want["sha_expected"]
            assert sha(z.to_host()) == want["sha_z"]
            for a in (d_counts, V, d_ltot, bgprob, expected, logp, z):
                a.free()
        for a in (counts, ltot, bg):
            a.free()
