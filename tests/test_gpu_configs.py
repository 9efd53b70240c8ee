"""BASELINE.json configs[3] and configs[4] at their real sizes, through the C ABI.

configs[3]  "Synthetic 100M x 200 bp, W=12, seqs sharded 8 x MI355X": ONE shard of 12.5M sequences on the one GPU of
            the test box (a shard is the most the reference's 32-bit position counters can take), checked
            through size-independent properties -- ltot, shard additivity (what the all-reduce relies on), the two
            independent emitters (direct atomics vs two-level partition) bin for bin, fused background totals.  The
            compiled reference's own results for this input live in tests/golden/synth_checksums.json and are compared by
            test_gpu_parity.py::test_device_generated_input_against_reference_checksums: every table for the 2M-sequence
            head of shard 0 (400 Mbp), and count table / ltot / background counters for the WHOLE shard 3 (2.65e9
            positions, just inside the reference's 32-bit counters; its background model overflows an int beyond 2^31
            bases, so V and z are not compared there).
configs[4]  "W=10 --strand PLUS, 1000 seed PWMs, EM-only stress" exactly as SURVEY.md 8(d) defines it: the PLUS count
            table of the 10M x 200 bp set, the 1000 highest-count k-mers (ties by ascending id) as seeds, initial PWM
            row 0.7 at the seed's base and 0.1 elsewhere, --em-threshold 0, 10 iterations = 1.05e10 evaluations.
            Sampled PWMs against the oracle: the library's default mode within BASELINE.json's 1e-5 relative of the
            fp64-accumulating restatement, the serial mode (what the CLI runs) bit for bit.
"""
import numpy as np
import pytest

import peng_motif_amd as pk
from oracle import oracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = pk.Context(0)
    # (this module's tests name the EM mode they run and put 1 back: the throughput mode is their base state; the
    # library's own default is 2, the bit-exact mode -- test_library_default_em_mode_is_the_bit_exact_one)
    c.set_option("em_fast", 1)
    yield c
    c.close()


def _count(ctx, seq0, n, L, W, both, impl=0, bg=False):
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


def test_config3_one_w12_shard_of_12_5M_sequences(ctx):
    W, L, n = 12, 200, 12_500_000
    shard = 3  # sequences [37.5M, 50M) of the 100M-sequence set
    seq0 = shard * n
    whole, lt, bg = _count(ctx, seq0, n, L, W, True, bg=True)
    assert lt == n * (L - W + 1)
    # fused background counters: totals follow from the geometry; the 1- and 2-mer marginals from the 3-mers
    assert int(bg[:4].sum()) == n * L and int(bg[4:20].sum()) == n * (L - 1) and int(bg[20:].sum()) == n * (L - 2)
    # shard additivity, bit for bit (the non-overlap rule never crosses a sequence boundary)
    a, la = _count(ctx, seq0, n // 2, L, W, True)
    acc = a.astype(np.uint64)
    del a
    b, lb = _count(ctx, seq0 + n // 2, n - n // 2, L, W, True)
    assert la + lb == lt
    acc += b
    del b
    assert np.array_equal(acc, whole.astype(np.uint64))
    del acc
    # counted + suppressed == visited, suppressed fraction tiny on random DNA (7.5e-6 at W = 10)
    counted = int(whole.astype(np.uint64).sum())
    assert 0 <= lt - counted < 1e-4 * lt
    # the direct-atomic emitter is an independent implementation of the same scan: every bin must agree
    direct, ld = _count(ctx, seq0, n, L, W, True, impl=1)
    assert ld == lt and np.array_equal(direct, whole)
    # plus strand at the same size: the emitters agree there too
    p2, lp = _count(ctx, seq0, n, L, W, False, impl=2)
    d2, _ = _count(ctx, seq0, n, L, W, False, impl=1)
    assert lp == lt and np.array_equal(p2, d2)
    assert int(p2.astype(np.uint64).sum()) >= counted  # the revcomp twin can only suppress more


def _top_seeds(counts, n):
    """the n highest-count k-mers, ties by ascending id"""
    order = np.lexsort((np.arange(counts.size), -counts.astype(np.int64)))[:n]
    return order


def _seed_pwms(seeds, W):
    pw = np.full((len(seeds), W, 4), 0.1, np.float32)
    for i, x in enumerate(seeds):
        for q in range(W):
            pw[i, q, (int(x) >> (2 * q)) & 3] = 0.7
    return pw


def test_config4_em_stress_1000_top_count_seeds_plus_table(ctx):
    W, L, n, K = 10, 200, 10_000_000, 2
    ctx.synth(1, 0, n, L, W)
    counts, ltot, bg = ctx.count_bg(False)  # PLUS: no mirror, no strand aggregation
    V = ctx.bg_model(bg, K)
    bgprob, expected, logp, z = ctx.pattern_stats(W, False, K, K, V, ltot, counts)
    c_host = counts.to_host()
    assert int(ltot.to_host()[0]) == n * (L - W + 1)
    seeds = _top_seeds(c_host, 1000)
    assert po.kmer_str(int(seeds[0]), W) == "GCTGAGTCAT"  # the planted motif has by far the highest plus-strand count
    pw0 = _seed_pwms(seeds, W)
    bgk_host = bgprob.to_host()[K]
    bg_k = pk.DeviceArray.from_host(ctx, bgk_host)
    sample = [0, 1, 2, 17, 255, 256, 500, 777, 998, 999]
    c64 = c_host.astype(np.uint64)

    # library default (em_fast = 1): 1e-5 relative against the fp64-accumulating oracle, iteration counts equal
    got, iters, change = ctx.em(W, pw0, counts, bg_k, 1e4, 0.0, 10)
    assert (iters == 10).all()
    assert np.allclose(got.sum(axis=2), 1.0, atol=1e-6)
    for i in sample:
        ref, it, _ = po.em(W, c64, bgk_host, pw0[i], 1e4, 0.0, 10, mode=1, final_norm=False)
        assert it == 10
        rel = np.abs(got[i].astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-3)
        assert rel.max() <= 1e-5, (i, rel.max())

    # serial mode (the CLI's): the reference's float32 sums in the reference's order, bit for bit
    ctx.set_option("em_fast", 2)
    try:
        ser, it2, ch2 = ctx.em(W, pw0, counts, bg_k, 1e4, 0.0, 10)
    finally:
        ctx.set_option("em_fast", 1)
    assert (it2 == 10).all()
    for i in sample:
        ref, it, ch = po.em(W, c64, bgk_host, pw0[i], 1e4, 0.0, 10, mode=0, final_norm=False)
        assert ser[i].tobytes() == ref.astype(np.float32).tobytes(), i
        assert np.float32(ch2[i]).view(np.uint32) == np.float32(ch).view(np.uint32)
    # ... and the scan that evaluates those sums (csrc/seqsum.h) agrees with the dependent-addition fold on all 1000
    ctx.set_option("em_fast", 2)
    try:
        for scan in (0, 1, 3):  # (`ser` above came from the default, 2: blocks evaluated ahead of their chain; 3: the same in two launches)
            ctx.test_em_generation(scan)
            dep, it3, ch3 = ctx.em(W, pw0, counts, bg_k, 1e4, 0.0, 10)
            assert dep.tobytes() == ser.tobytes() and it3.tolist() == it2.tolist() and ch3.tobytes() == ch2.tobytes(), scan
    finally:
        ctx.test_em_generation(2)
        ctx.set_option("em_fast", 1)
    # The two modes differ by the REFERENCE's own float32 summation error, which grows with the table: 262144 serial
    # float32 additions per cell over 1.9e9 counted windows leave up to ~1e-3 absolute on a PWM entry here (SURVEY.md A.7
    # measured 1.9e-5 on a 1M-sequence set) -- the fp64-tree mode is the more accurate of the two.
    assert np.abs(ser.astype(np.float64) - got).max() <= 3e-3


def test_serial_em_at_w12_bit_exact_against_oracle(ctx):
    """The serial mode at W = 12: 48 cells of 4^11 = 4.2M weights each (1024 blocks of the scan per cell), three PWMs,
    three iterations, against the oracle's left-to-right float32 sums -- bit for bit, on the count table of a 100k-sequence
    synthetic set (most of the 16.7M k-mers have count 0: long stretches of zero terms in every cell)."""
    W, L, n, K = 12, 200, 100_000, 2
    ctx.synth(1, 0, n, L, W)
    counts, ltot, bg = ctx.count_bg(True)
    ctx.mirror(W, counts)
    V = ctx.bg_model(bg, K)
    bgprob, expected, logp, z = ctx.pattern_stats(W, True, K, K, V, ltot, counts)
    c_host = counts.to_host()
    bgk_host = bgprob.to_host()[K]
    seeds = _top_seeds(c_host, 3)
    pw0 = _seed_pwms(seeds, W)
    rng = np.random.default_rng(12)
    pw0[2] = rng.dirichlet(np.ones(4), size=W).astype(np.float32)
    bg_k = pk.DeviceArray.from_host(ctx, bgk_host)
    ctx.set_option("em_fast", 2)
    try:
        got, iters, change = ctx.em(W, pw0, counts, bg_k, 1e4, 0.0, 3)
    finally:
        ctx.set_option("em_fast", 1)
    c64 = c_host.astype(np.uint64)
    for i in range(3):
        ref, it, ch = po.em(W, c64, bgk_host, pw0[i], 1e4, 0.0, 3, mode=0, final_norm=False)
        assert iters[i] == it == 3
        assert got[i].tobytes() == ref.astype(np.float32).tobytes(), i
        assert np.float32(change[i]).view(np.uint32) == np.float32(ch).view(np.uint32)
