"""BASELINE-size checks (configs[2]: synthetic 10M x 200 bp, W=10) through size-independent properties,
since the oracle cannot finish these sizes in seconds:

  * ltot == n_seq * (L - W + 1) (no N in the synthetic set)
  * shard additivity: count(first half) + count(second half) == count(whole), bit for bit
    (the non-overlap rule never crosses a sequence boundary; this is what the multi-GPU all-reduce relies on)
  * the two independent emitters (direct global atomics vs partitioned LDS histograms) agree bin for bin
  * sum of canonical bins + suppressed windows == ltot, with the suppressed fraction in the range the oracle
    shows on a 200k-sequence sample of the same generator
  * mirror symmetry and idempotence; fused background counters == standalone K1b == closed form totals
  * the sweep is a pure function: bit-identical z on a second run; EM is deterministic run to run
"""
import numpy as np
import pytest

import peng_motif_amd as pk
from oracle import oracle as po

pytestmark = pytest.mark.gpu

W, L = 10, 200


@pytest.fixture(scope="module")
def ctx():
    c = pk.Context(0)
    yield c
    c.close()


def count_range(ctx, seq0, n, both, impl=0, bg=False):
    ctx.synth(1, seq0, n, L, W)
    ctx.set_option("count_impl", impl)
    try:
        if bg:
            c, lt, b = ctx.count_bg(both)
            return c.to_host(), int(lt.to_host()[0]), b.to_host()
        c, lt = ctx.count(both)
        return c.to_host(), int(lt.to_host()[0])
    finally:
        ctx.set_option("count_impl", 0)


def test_full_size_shard_additivity_and_ltot(ctx):
    n = 10_000_000
    whole, lt, bg = count_range(ctx, 0, n, True, bg=True)
    assert lt == n * (L - W + 1)
    a, la = count_range(ctx, 0, n // 2, True)
    b, lb = count_range(ctx, n // 2, n // 2, True)
    assert la + lb == lt
    assert np.array_equal(a.astype(np.uint64) + b.astype(np.uint64), whole.astype(np.uint64))
    # canonical bins only before the mirror; counted + suppressed == visited
    counted = int(whole.astype(np.uint64).sum())
    suppressed = lt - counted
    codes, offs = po.synth(1, 0, 200_000, L)
    want, lt_s = po.count(codes, offs, W, True)
    # oracle counts are mirrored: canonical sum = (sum + palindromes) / 2; a palindrome is fixed by its first half
    pal_ids = [h | (po.revcomp(h, 5) << 10) for h in range(4 ** 5)]
    assert all(po.revcomp(x, W) == x for x in pal_ids[:50])
    canon_sum = (int(want.sum()) + int(want[pal_ids].sum())) // 2
    frac_sample = (lt_s - canon_sum) / lt_s
    frac_full = suppressed / lt
    assert 0 < frac_full < 1e-3 and abs(frac_full - frac_sample) < 0.25 * frac_sample + 1e-7
    # background counters: totals follow from the sequence geometry
    assert int(bg[:4].sum()) == n * L and int(bg[4:20].sum()) == n * (L - 1) and int(bg[20:].sum()) == n * (L - 2)


def test_direct_and_partitioned_emitters_agree_at_scale(ctx):
    n = 2_000_000
    d, ld = count_range(ctx, 123_456, n, True, impl=1)
    p, lp, bgp = count_range(ctx, 123_456, n, True, impl=2, bg=True)
    assert ld == lp and np.array_equal(d, p)
    d2, _ = count_range(ctx, 123_456, n, False, impl=1)
    p2, _ = count_range(ctx, 123_456, n, False, impl=2)
    assert np.array_equal(d2, p2)
    # plus-strand counts fold onto the canonical ones except where the revcomp twin suppressed a window
    assert int(p2.astype(np.uint64).sum()) >= int(p.astype(np.uint64).sum())
    standalone = ctx.bg_count().to_host()
    assert np.array_equal(standalone, bgp)


def test_mirror_sweep_and_em_are_deterministic(ctx):
    n = 1_000_000
    ctx.synth(1, 0, n, L, W)
    counts, ltot, bg = ctx.count_bg(True)
    ctx.mirror(W, counts)
    c = counts.to_host()
    idx = np.random.default_rng(0).integers(0, 4 ** W, size=5000)
    for x in idx.tolist():
        assert c[x] == c[po.revcomp(x, W)]
    ctx.mirror(W, counts)
    assert np.array_equal(counts.to_host(), c)  # idempotent
    V = ctx.bg_model(bg, 2)
    b1, e1, l1, z1 = ctx.pattern_stats(W, True, 2, 2, V, ltot, counts)
    b2, e2, l2, z2 = ctx.pattern_stats(W, True, 2, 2, V, ltot, counts)
    assert z1.to_host().tobytes() == z2.to_host().tobytes() and b1.to_host().tobytes() == b2.to_host().tobytes()
    bgp = b1.to_host()
    assert abs(float(bgp[2].sum(dtype=np.float64)) - 2.0) < 1e-3  # strand-aggregated probabilities sum to ~2
    assert abs(float(e1.to_host().sum(dtype=np.float64)) / (2.0 * n * (L - W + 1)) - 1.0) < 1e-3
    # the planted motif is the top z-score (either strand)
    top = int(np.argmax(z1.to_host()))
    assert po.kmer_str(top, W) in ("GCTGAGTCAT", "ATGACTCAGC")
    # EM: same input twice -> same bits (fixed-order fp64 reduction)
    pw = np.full((8, W, 4), 0.1, np.float32)
    for i in range(8):
        for q in range(W):
            pw[i, q, ((top + 977 * i) >> (2 * q)) & 3] = 0.7
    bg2 = pk.DeviceArray.from_host(ctx, bgp[2])
    got = {}
    try:
        for mode in (2, 1):  # the library's default (serial float32 sums, bit-exact) and the throughput mode (fixed-order fp64 tree)
            ctx.set_option("em_fast", mode)
            r1, it1, _ = ctx.em(W, pw, counts, bg2, 1e4, 0.08, 10)
            r2, it2, _ = ctx.em(W, pw, counts, bg2, 1e4, 0.08, 10)
            assert r1.tobytes() == r2.tobytes() and np.array_equal(it1, it2)
            assert np.allclose(r1.sum(axis=2), 1.0, atol=1e-6)
            got[mode] = (r1, it1)
    finally:
        ctx.set_option("em_fast", 2)
    # (the two modes may stop a PWM one iteration apart when `change` lands next to the threshold -- why the bit-exact mode
    # is the default; where they ran the same number of iterations the PWMs agree to the float32 sums' own error)
    same = got[1][1] == got[2][1]
    assert same.sum() >= 6 and np.abs(got[1][1] - got[2][1]).max() <= 1
    assert np.abs(got[1][0][same].astype(np.float64) - got[2][0][same]).max() <= 3e-3
