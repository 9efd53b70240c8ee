"""GPU parity: the HIP path, called through the C ABI (libpengk.so), against the oracle on the same
inputs and against the committed golden vectors of the compiled reference.

Bars (SURVEY.md 8c / BASELINE.md): integer tables bit-exact; bgprob / expected / z bit-exact
(float32 bit patterns); log-p equal up to 1 float ulp (device libm log vs glibc); IUPAC sums
bit-exact; EM PWMs <= 1e-6 absolute vs the fp64-accumulating oracle and <= 1e-4 absolute vs the
reference's own float32 serial sums, iteration counts equal.
"""
import os

import numpy as np
import pytest

import peng_motif_amd as pk
from oracle import oracle as po

pytestmark = pytest.mark.gpu

CASES = ["mafk100_w8_both", "mafk100_w8_plus", "mafk100_w6_both", "torture_w6_both", "torture_w6_plus",
         "torture_w4_both", "torture_w8_plus", "mafk_w10_both", "mafk_w10_plus", "torture_w2_both", "mafk100_w2_plus"]


@pytest.fixture(scope="module")
def ctx():
    c = pk.Context(0)
    # (this module's tests name the EM mode they run and put 1 back: the throughput mode is their base state; the
    # library's own default is 2, the bit-exact mode -- test_library_default_em_mode_is_the_bit_exact_one)
    c.set_option("em_fast", 1)
    yield c
    c.close()


def bits_equal(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    return a.shape == b.shape and a.tobytes() == b.tobytes()


def ulp_diff(a, b):
    """max difference in float32 ulps, treating equal infinities as 0."""
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    d = np.abs(ia - ib)
    d[same] = 0
    return int(d.max()) if d.size else 0


_cpu = {}


def cpu_pipeline(golden_dir, name):
    if name in _cpu:
        return _cpu[name]
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    W, both, K = int(g["W"]), bool(g["both"]), int(g["K"])
    codes, offs = po.read_fasta(os.path.join(golden_dir, str(g["fasta"])))
    n = po.bg_counts(codes, offs, 2)
    V = po.bg_V(n, 2)
    counts, ltot = po.count(codes, offs, W, both)
    bgp = [po.bgprob(W, k, V, both) for k in range(K + 1)]
    e, lp, z = po.stats(W, counts, bgp[K], ltot)
    r = dict(g=g, W=W, both=both, K=K, codes=codes, offs=offs, bgcounts=n, V=V, counts=counts, ltot=ltot, bgp=bgp,
             expected=e, logp=lp, z=z)
    _cpu[name] = r
    return r


def gpu_tables(ctx, r, item_windows=0):
    """pack -> upload -> count -> mirror -> bg model -> stats; returns host copies + device handles."""
    W, both, K = r["W"], r["both"], r["K"]
    p = pk.Packed(r["codes"], r["offs"], W, item_windows)
    ctx.upload(p)
    counts, ltot = ctx.count(both)
    if both:
        ctx.mirror(W, counts)
    bg = ctx.to_device(p.bg_counts.astype(np.uint64))
    V = ctx.bg_model(bg, 2)
    bgprob, expected, logp, z = ctx.pattern_stats(W, both, K, K, V, ltot, counts)
    return dict(p=p, counts=counts, ltot=ltot, V=V, bgprob=bgprob, expected=expected, logp=logp, z=z)


@pytest.mark.parametrize("name", CASES)
def test_count_and_sweep_bit_exact(ctx, golden_dir, name):
    r = cpu_pipeline(golden_dir, name)
    d = gpu_tables(ctx, r)
    assert int(d["ltot"].to_host()[0]) == r["ltot"] == int(r["g"]["ltot"])
    assert bits_equal(d["counts"].to_host().astype(np.uint64), r["counts"])
    assert bits_equal(d["V"].to_host(), r["V"])
    bgp = d["bgprob"].to_host()
    for k in range(r["K"] + 1):
        assert bits_equal(bgp[k], r["bgp"][k]), "bgp%d" % k
    assert bits_equal(d["expected"].to_host(), r["expected"])
    assert bits_equal(d["z"].to_host(), r["z"])
    assert ulp_diff(d["logp"].to_host(), r["logp"]) <= 1
    # the seeds the reference selects follow from bit-identical z: run the host selection on the GPU tables
    seeds = po.select(r["W"], d["z"].to_host(), d["counts"].to_host().astype(np.uint64), 10.0, 3, not r["both"], True)
    assert bits_equal(seeds, r["g"]["seeds"])


@pytest.mark.parametrize("k", [0, 1])
@pytest.mark.parametrize("name", ["mafk100_w8_both", "torture_w6_plus"])
def test_sweep_lower_background_orders(ctx, golden_dir, name, k):
    """--bg-model-order 0 / 1: z-scores use order k while all orders 0..max_k are tabulated (EM uses max_k)."""
    r = cpu_pipeline(golden_dir, name)
    W, both = r["W"], r["both"]
    p = pk.Packed(r["codes"], r["offs"], W)
    ctx.upload(p)
    counts, ltot = ctx.count(both)
    if both:
        ctx.mirror(W, counts)
    V = ctx.bg_model(ctx.to_device(p.bg_counts.astype(np.uint64)), 2)
    bgprob, expected, logp, z = ctx.pattern_stats(W, both, k, 2, V, ltot, counts)
    e, lp, zz = po.stats(W, r["counts"], r["bgp"][k], r["ltot"])
    bgp = bgprob.to_host()
    for o in range(3):
        assert bits_equal(bgp[o], r["bgp"][o])
    assert bits_equal(expected.to_host(), e) and bits_equal(z.to_host(), zz)
    assert ulp_diff(logp.to_host(), lp) <= 1


@pytest.mark.parametrize("name", ["torture_w6_both", "torture_w6_plus", "torture_w8_plus", "mafk100_w8_both"])
def test_count_with_split_items_and_deferred_fixup(ctx, golden_dir, name):
    """item_windows = 64 splits every run longer than 64 windows: poly-A / (AT)n / tandem repeats force
    the prologue to fail certification and exercise count_fixup_kernel."""
    r = cpu_pipeline(golden_dir, name)
    p = pk.Packed(r["codes"], r["offs"], r["W"], 64)
    ctx.upload(p)
    counts, ltot = ctx.count(r["both"])
    deferred = ctx.info("deferred_items")
    if name.startswith("torture"):
        assert deferred > 0  # the fallback really ran
    if r["both"]:
        ctx.mirror(r["W"], counts)
    assert int(ltot.to_host()[0]) == r["ltot"]
    assert bits_equal(counts.to_host().astype(np.uint64), r["counts"])


@pytest.mark.parametrize("impl", [1, 2])
@pytest.mark.parametrize("name", ["mafk100_w8_both", "torture_w8_plus", "mafk_w10_both", "mafk_w10_plus"])
def test_count_both_implementations(ctx, golden_dir, name, impl):
    """direct global atomics (1) and partitioned LDS histograms (2) give the same table."""
    r = cpu_pipeline(golden_dir, name)
    p = pk.Packed(r["codes"], r["offs"], r["W"], 64 if "torture" in name else 0)
    ctx.upload(p)
    ctx.set_option("count_impl", impl)
    try:
        counts, ltot = ctx.count(r["both"])
        if r["both"]:
            ctx.mirror(r["W"], counts)
        assert int(ltot.to_host()[0]) == r["ltot"]
        assert bits_equal(counts.to_host().astype(np.uint64), r["counts"])
    finally:
        ctx.set_option("count_impl", 0)


@pytest.mark.parametrize("both", [True, False])
def test_w12_two_level_partition(ctx, both):
    """W = 12: scan -> 32 coarse buckets (32-bit keys) -> 16 fine buckets (16-bit keys) -> 512 LDS histograms;
    against the oracle on a small set and against the direct-atomic emitter on a larger one, incl. fused K1b,
    split items and an overflowing key buffer."""
    W = 12
    codes, offs = po.synth(5, 0, 4000, 150)
    want, ltot = po.count(codes, offs, W, both)
    for M, cap in ((0, 0), (64, 0), (0, 64)):
        p = pk.Packed(codes, offs, W, M)
        ctx.upload(p)
        ctx.set_option("count_impl", 2)
        ctx.set_option("key_cap_override", cap)
        try:
            counts, lt, bg = ctx.count_bg(both)
            if both:
                ctx.mirror(W, counts)
            assert int(lt.to_host()[0]) == ltot
            assert np.array_equal(counts.to_host().astype(np.uint64), want), (M, cap)
            assert np.array_equal(bg.to_host().astype(np.int64), po.bg_counts(codes, offs, 2))
        finally:
            ctx.set_option("count_impl", 0)
            ctx.set_option("key_cap_override", 0)
    n = 400_000
    ctx.synth(1, 77, n, 200, W)
    ctx.set_option("count_impl", 1)
    d, ld = ctx.count(both)
    d = d.to_host()
    ctx.set_option("count_impl", 2)
    try:
        q, lq = ctx.count(both)
        assert int(ld.to_host()[0]) == int(lq.to_host()[0]) == n * (200 - W + 1)
        assert np.array_equal(d, q.to_host())
    finally:
        ctx.set_option("count_impl", 0)


def _same_bucket_sets(W, L=120):
    """Inputs that put many lanes of one wave on ONE partition bucket in the same step: the lanes of a wave hold
    consecutive scan items, so adjacent identical sequences make them emit identical keys."""
    rng = np.random.default_rng(12)
    unit13 = rng.integers(1, 5, size=13).astype(np.uint8)  # period 13 > W - 1: every window is counted
    rep13 = np.tile(unit13, L // 13 + 1)[:L]
    poly_t = np.full(L, 4, np.uint8)
    one = rng.integers(1, 5, size=L).astype(np.uint8)
    sets = {}
    sets["period13_x4000"] = np.tile(rep13, 4000)
    sets["polyT_x4000"] = np.tile(poly_t, 4000)
    sets["identical_random_x4000"] = np.tile(one, 4000)
    mixed = []
    for w in range(60):  # 64 sequences per wave: 40 copies of one sequence + 24 random ones
        mixed.append(np.tile(rng.integers(1, 5, size=L).astype(np.uint8), 40))
        mixed.append(rng.integers(1, 5, size=24 * L).astype(np.uint8))
    sets["40_identical_24_random_per_wave"] = np.concatenate(mixed)
    return {k: (v, np.arange(len(v) // L + 1, dtype=np.int64) * L) for k, v in sets.items()}


@pytest.mark.parametrize("both", [True, False])
@pytest.mark.parametrize("W", [12, 10, 8])
def test_partitioned_count_many_lanes_on_one_bucket(ctx, W, both):
    """All 64 lanes of a wave appending to one (wave, bucket) ring in a single step must not overwrite unflushed
    ring entries (W = 12 level 1: 64-entry rings, groups of 32 -- the append runs in two half-waves; 16-bit rings:
    128 entries, groups of 64).  ltot alone would not notice: the bins are compared with the oracle."""
    for name, (codes, offs) in _same_bucket_sets(W).items():
        want, ltot = po.count(codes, offs, W, both)
        p = pk.Packed(codes, offs, W)
        ctx.upload(p)
        ctx.set_option("count_impl", 2)
        try:
            counts, lt = ctx.count(both)
            if both:
                ctx.mirror(W, counts)
            assert int(lt.to_host()[0]) == ltot, name
            got = counts.to_host().astype(np.uint64)
            assert np.array_equal(got, want), "%s: %d mismatching bins" % (name, int((got != want).sum()))
        finally:
            ctx.set_option("count_impl", 0)


def test_partitioned_count_survives_a_full_bucket_region(ctx):
    """Skew: every counted window lands in few buckets and the key-buffer hint is far too small, so bucket
    regions run full and the overflow path (direct atomics) must keep the table exact."""
    W = 10
    rng = np.random.default_rng(3)
    unit = rng.integers(1, 5, size=13).astype(np.uint8)  # period 13 > W: every window is counted, 13 distinct ids
    L = 500
    codes = np.tile(np.tile(unit, L // 13 + 1)[:L], 4000)
    offs = np.arange(4001, dtype=np.int64) * L
    want, ltot = po.count(codes, offs, W, True)
    p = pk.Packed(codes, offs, W)
    ctx.upload(p)
    ctx.set_option("count_impl", 2)
    ctx.set_option("key_cap_override", 128)  # two groups per (wave, bucket) slice, then the slices are full
    try:
        counts, lt = ctx.count(True)
        ctx.mirror(W, counts)
        assert int(lt.to_host()[0]) == ltot
        assert np.array_equal(counts.to_host().astype(np.uint64), want)
    finally:
        ctx.set_option("count_impl", 0)
        ctx.set_option("key_cap_override", 0)


@pytest.mark.parametrize("both", [True, False])
def test_partitioned_count_w14_three_levels(ctx, both):
    """W = 14 (2^28 bins = 2^13 LDS histograms): three partition levels -- scan -> 16 buckets of 32-bit keys -> 16 each
    of 32-bit keys -> 32 each of 16-bit keys -> pass B.  Against the direct (one atomic per window) emitter, which
    test_count_low_complexity_long_sequences pins to the oracle at W = 14: random sequences, the many-lanes-on-one-bucket
    sets (identical sequences, a period-13 repeat, poly-T), and slices forced to overflow at every level."""
    W = 14
    rng = np.random.default_rng(14)
    n, L = 30000, 180
    codes = rng.integers(1, 5, size=n * L).astype(np.uint8)
    codes[rng.integers(0, codes.size, 200)] = 0  # a few N: more than one run per sequence, items that start mid-sequence
    cases = {"random_30000x180": (codes, np.arange(n + 1, dtype=np.int64) * L, 0)}
    for name, (c, o) in _same_bucket_sets(W).items():
        cases[name] = (c, o, 0)
    cases["random_overflowing_slices"] = (codes, np.arange(n + 1, dtype=np.int64) * L, 64)
    for name, (c, o, cap) in cases.items():
        ctx.upload(pk.Packed(c, o, W))
        ctx.set_option("count_impl", 1)
        try:
            want, lt1 = ctx.count(both)
            want, lt1 = want.to_host().copy(), int(lt1.to_host()[0])
            ctx.set_option("count_impl", 2)
            ctx.set_option("key_cap_override", cap)
            got, lt2 = ctx.count(both)
            assert int(lt2.to_host()[0]) == lt1, name
            got = got.to_host()
            assert np.array_equal(got, want), "%s: %d mismatching bins" % (name, int((got != want).sum()))
            assert int(got.sum()) > 0
        finally:
            ctx.set_option("count_impl", 0)
            ctx.set_option("key_cap_override", 0)


def low_complexity_set(seed, n, L):
    rng = np.random.default_rng(seed)
    seqs = []
    for i in range(n):
        kind = i % 6
        if kind == 0:
            s = np.full(L, 1 + (i // 6) % 4, np.uint8)  # homopolymer
        elif kind == 1:
            unit = rng.integers(1, 5, size=int(rng.integers(2, 7)))
            s = np.tile(unit, L // len(unit) + 1)[:L].astype(np.uint8)  # short tandem repeat
        elif kind == 2:
            half = rng.integers(1, 5, size=L // 2)
            s = np.concatenate([half, (5 - half[::-1])])[:L].astype(np.uint8)  # hairpin (revcomp palindrome)
        elif kind == 3:
            s = rng.integers(1, 5, size=L).astype(np.uint8)
            s[rng.integers(0, L, size=3)] = 0  # a few N
        elif kind == 4:
            unit = rng.integers(1, 5, size=11)
            s = np.tile(unit, L // 11 + 1)[:L].astype(np.uint8)  # period > W-1 for W <= 10
        else:
            s = rng.integers(1, 5, size=L).astype(np.uint8)
        seqs.append(s)
    codes = np.concatenate(seqs)
    offs = np.arange(n + 1, dtype=np.int64) * L
    return codes, offs


@pytest.mark.parametrize("W,both,L,M", [(6, True, 700, 64), (8, False, 1500, 64), (10, True, 1000, 100), (12, False, 520, 256),
                                        (4, True, 300, 64), (14, True, 400, 64)])
def test_count_low_complexity_long_sequences(ctx, W, both, L, M):
    codes, offs = low_complexity_set(W * 100 + L, 48, L)
    want, ltot = po.count(codes, offs, W, both)
    p = pk.Packed(codes, offs, W, M)
    ctx.upload(p)
    counts, lt = ctx.count(both)
    if both:
        ctx.mirror(W, counts)
    assert int(lt.to_host()[0]) == ltot
    got = counts.to_host().astype(np.uint64)
    assert np.array_equal(got, want), "mismatching bins: %d" % int((got != want).sum())


def test_count_empty_and_short_inputs(ctx):
    # no sequence reaches W bases: no items, zero table
    codes = np.array([1, 2, 3, 1, 2, 0, 1, 2, 3, 4, 1], np.uint8)
    offs = np.array([0, 3, 11], np.int64)
    p = pk.Packed(codes, offs, 6)
    assert len(p.items) == 0
    ctx.upload(p)
    counts, lt = ctx.count(True)
    assert int(lt.to_host()[0]) == 0 and int(counts.to_host().sum()) == 0
    with pytest.raises(pk.PengkError) as e:  # bg recount needs whole-sequence runs
        ctx.bg_count()
    assert e.value.code == pk.ERR_UNSUPPORTED


def test_device_bg_count_matches_packer_and_oracle(ctx, golden_dir):
    codes, offs = po.read_fasta(os.path.join(golden_dir, "MafK.fasta"))
    for M in (64, 0):
        p = pk.Packed(codes, offs, 10, M)
        assert p.all_whole == 1
        ctx.upload(p)
        got = ctx.bg_count().to_host().astype(np.int64)
        assert np.array_equal(got, p.bg_counts)
        assert np.array_equal(got, po.bg_counts(codes, offs, 2))


@pytest.mark.parametrize("impl", [1, 2])
@pytest.mark.parametrize("W,M", [(10, 64), (10, 0), (8, 64), (6, 0), (12, 0), (12, 64)])
def test_fused_bg_count(ctx, golden_dir, W, M, impl):
    """pengk_count_bg: the 3-mer bins collected inside the count scan == the packer's / oracle's counts,
    and the count table is unchanged.  M = 64 splits runs (continuing items must not recount their prologue)."""
    if impl == 2 and W not in (8, 10, 12, 14):
        pytest.skip("partitioned count is built for W = 8 .. 14")
    codes, offs = po.read_fasta(os.path.join(golden_dir, "MafK.fasta"))
    codes, offs = codes[:offs[600]], offs[:601]
    p = pk.Packed(codes, offs, W, M)
    ctx.upload(p)
    ctx.set_option("count_impl", impl)
    try:
        counts, lt, bg = ctx.count_bg(False)
        want, ltot = po.count(codes, offs, W, False)
        assert int(lt.to_host()[0]) == ltot
        assert np.array_equal(counts.to_host().astype(np.uint64), want)
        assert np.array_equal(bg.to_host().astype(np.int64), po.bg_counts(codes, offs, 2))
    finally:
        ctx.set_option("count_impl", 0)


def test_fused_bg_count_with_deferred_items(ctx):
    """low-complexity runs split into items: deferred items still contribute their bases to the fused bins."""
    codes, offs = low_complexity_set(11, 48, 900)
    keep = [i for i in range(48) if i % 6 != 3]  # drop the sequences with N (bg recount needs whole runs)
    codes = np.concatenate([codes[offs[i]:offs[i + 1]] for i in keep])
    offs = np.arange(len(keep) + 1, dtype=np.int64) * 900
    p = pk.Packed(codes, offs, 10, 64)
    assert p.all_whole == 1
    ctx.upload(p)
    counts, lt, bg = ctx.count_bg(True)
    assert ctx.info("deferred_items") > 0
    ctx.mirror(10, counts)
    want, ltot = po.count(codes, offs, 10, True)
    assert np.array_equal(counts.to_host().astype(np.uint64), want)
    assert np.array_equal(bg.to_host().astype(np.int64), po.bg_counts(codes, offs, 2))


def _synth_cases():
    import json
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "synth_checksums.json")
    return json.load(open(path))


@pytest.mark.parametrize("mirrored", [True, False])
@pytest.mark.parametrize("W", [12, 14])
def test_sweep_twin_tiles_equal_the_per_pattern_kernel(ctx, W, mirrored):
    """Both strands, W >= 12: stats_pair_kernel evaluates a pattern and its reverse complement once (a tile of 64 x 64
    patterns and the tile of the twins, handed over through LDS; csrc/stats.hip) -- every table must carry the bits of the
    one-thread-per-pattern kernel (option sweep_pairs = 0; that kernel is the one the goldens and the reference checksums
    pin at W <= 10 and, through sha256, at W = 12): all background orders, expected, log-p, z, for k = 0 / 1 / 2, on a
    mirrored count table and on an un-mirrored one (twins with different counts: log-p and z are then worked out for
    both), palindromic patterns and the 64 tiles that are their own twins included (the whole table is compared)."""
    ctx.synth(5, 0, 60000 if W == 12 else 200000, 200, W)
    counts, ltot, bg = ctx.count_bg(True)
    if mirrored:
        ctx.mirror(W, counts)
    V = ctx.bg_model(bg, 2)
    try:
        for k, max_k in ((2, 2), (0, 2), (1, 1)):
            ctx.set_option("sweep_pairs", 0)
            want = [a.to_host().copy() for a in ctx.pattern_stats(W, True, k, max_k, V, ltot, counts)]
            ctx.set_option("sweep_pairs", 1)
            got = [a.to_host() for a in ctx.pattern_stats(W, True, k, max_k, V, ltot, counts)]
            for name, g, w in zip(("bgprob", "expected", "logp", "z"), got, want):
                assert g.tobytes() == w.tobytes(), (W, mirrored, k, max_k, name, int((g.view(np.uint32) != w.view(np.uint32)).sum()))
        c = counts.to_host()
        if not mirrored:
            rc = np.array([po.revcomp(x, W) for x in range(0, 4 ** W, 4 ** W // 4096 + 1)])
            assert (c[::4 ** W // 4096 + 1] != c[rc]).any()  # (the table really is un-mirrored)
    finally:
        ctx.set_option("sweep_pairs", 1)


@pytest.mark.parametrize("case", _synth_cases(), ids=lambda c: "W%d_%s_%d" % (c["W"], c["strand"], c["n_seq"]))
def test_device_generated_input_against_reference_checksums(ctx, case):
    """Sequences generated ON THE DEVICE (pengk_synth_sequences) -> fused count -> bg model -> sweep, compared
    with sha256 checksums of the compiled reference's tables for the same synthetic set
    (tests/golden/make_synth_golden.py): tens of Mbp at W = 10 and W = 12, beyond the per-element fixtures."""
    import hashlib
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
    W, both = case["W"], case["strand"] == "BOTH"
    ctx.synth(case["seed"], case["seq0"], case["n_seq"], case["L"], W)
    counts, ltot, bg = ctx.count_bg(both)
    if both:
        ctx.mirror(W, counts)
    assert int(ltot.to_host()[0]) == case["ltot"]
    assert bg.to_host().astype(np.int64).tolist() == case["bgcounts"]
    assert sha(counts.to_host()) == case["sha_counts_u32"]
    if case["n_seq"] * case["L"] >= 2 ** 31:
        # Beyond 2^31 bases the reference's background model sums its base counts in an `int`
        # (src/shared/BackgroundModel.cpp:497) and its V, expected counts and z-scores are those of the wrapped sum; the
        # count table, ltot and the 84 counters above are what it still computes correctly -- and they match.
        return
    V = ctx.bg_model(bg, 2)
    assert sha(V.to_host()) == case["sha_V"]
    bgprob, expected, logp, z = ctx.pattern_stats(W, both, 2, 2, V, ltot, counts)
    assert sha(bgprob.to_host()[2]) == case["sha_bgp2"]
    assert sha(expected.to_host()) == case["sha_expected"]
    zz = z.to_host()
    assert sha(zz) == case["sha_z"]
    seeds = po.select(W, zz, counts.to_host().astype(np.uint64), 10.0, 3, not both, True)
    assert seeds[:40].tolist() == case["seeds"]


def test_synthetic_generator_matches_cpu(ctx):
    """pengk_synth_sequences == the counter-based generator of SURVEY.md 8d (oracle po_synth)."""
    n, L, W = 3000, 200, 10
    codes, offs = po.synth(1, 1234, n, L)
    p = pk.Packed(codes, offs, W)
    words, items, nw, ni = ctx.synth(1, 1234, n, L, W)
    assert nw == len(p.words) and ni == len(p.items)
    assert np.array_equal(words.to_host(), p.words)
    assert np.array_equal(items.to_host()[:ni], p.items)
    counts, lt = ctx.count(False)
    want, ltot = po.count(codes, offs, W, False)
    assert int(lt.to_host()[0]) == ltot
    assert np.array_equal(counts.to_host().astype(np.uint64), want)
    assert np.array_equal(ctx.bg_count().to_host().astype(np.int64), po.bg_counts(codes, offs, 2))


@pytest.mark.parametrize("name", ["mafk100_w8_both", "torture_w8_plus", "torture_w6_both", "mafk_w10_both", "mafk_w10_plus"])
def test_iupac_aggregation_bit_exact(ctx, golden_dir, name):
    r = cpu_pipeline(golden_dir, name)
    g = r["g"]
    d = gpu_tables(ctx, r)
    ids = g["iupac_ids"]
    bgp_k = pk.DeviceArray.from_host(ctx, d["bgprob"].to_host()[r["K"]])
    out = ctx.iupac_aggregate(r["W"], r["both"], ids, d["counts"], bgp_k, d["expected"])
    assert np.array_equal(out["sites"], g["iupac_sites"])
    assert np.array_equal(out["sites"], g["iupac_cc"])
    fb = g["iupac_fbits"]
    for j, f in enumerate(["bg_p", "expected", "zscore", "log_pvalue"]):
        assert np.array_equal(out[f].view(np.uint32), fb[:, j]), f


def test_library_default_em_mode_is_the_bit_exact_one(golden_dir):
    """A context nobody set an option on runs pengk_em in the serial mode: the compiled reference's PWMs (golden pwm_post
    is what the reference holds after its constructor's extra normalisation), iteration counts and the oracle's
    float32 `change`, bit for bit -- a caller that binds the ABI gets the reference's results without asking."""
    r = cpu_pipeline(golden_dir, "mafk_w10_both")
    g, W, K = r["g"], r["W"], r["K"]
    c = pk.Context(0)
    try:
        d = gpu_tables(c, r)
        bg_k = pk.DeviceArray.from_host(c, d["bgprob"].to_host()[K])
        pw, iters, change = c.em(W, g["pwm_pre"], d["counts"], bg_k, 1e4, 0.08, 10)
    finally:
        c.close()
    assert len(g["pwm_ids"]) > 0
    for i in range(len(g["pwm_ids"])):
        ref, it, ch = po.em(W, r["counts"], r["bgp"][K], g["pwm_pre"][i], 1e4, 0.08, 10, mode=0, final_norm=False)
        assert iters[i] == it == int(g["em_iters"][i])
        assert pw[i].tobytes() == ref.astype(np.float32).tobytes()
        assert np.float32(change[i]).view(np.uint32) == np.float32(ch).view(np.uint32)
        fin = pw[i] / pw[i].sum(axis=1, keepdims=True, dtype=np.float32)
        assert bits_equal(fin.astype(np.float32), g["pwm_post"][i].astype(np.float32))


@pytest.mark.parametrize("name", ["mafk100_w8_both", "mafk100_w8_plus", "torture_w6_plus", "mafk_w10_both", "mafk_w10_plus"])
def test_em_against_oracle_and_reference(ctx, golden_dir, name):
    r = cpu_pipeline(golden_dir, name)
    g = r["g"]
    W, K = r["W"], r["K"]
    if len(g["pwm_ids"]) == 0:
        pytest.skip("no PWMs in this case")
    d = gpu_tables(ctx, r)
    bg_k = pk.DeviceArray.from_host(ctx, d["bgprob"].to_host()[K])
    pw, iters, change = ctx.em(W, g["pwm_pre"], d["counts"], bg_k, 1e4, 0.08, 10)
    dev64 = 0.0
    devref = 0.0
    for i in range(len(g["pwm_ids"])):
        p1, it1, ch1 = po.em(W, r["counts"], r["bgp"][K], g["pwm_pre"][i], 1e4, 0.08, 10, mode=1, final_norm=False)
        assert iters[i] == it1 == int(g["em_iters"][i])
        dev64 = max(dev64, float(np.abs(pw[i].astype(np.float64) - p1).max()))
        # the reference's constructor normalises once more (src/iupac_pattern.cpp:61)
        fin = pw[i] / pw[i].sum(axis=1, keepdims=True, dtype=np.float32)
        devref = max(devref, float(np.abs(fin.astype(np.float64) - g["pwm_post"][i]).max()))
        assert abs(float(change[i]) - ch1) <= 1e-5
    assert dev64 <= 1e-6, dev64     # vs the fp64-accumulating restatement
    assert devref <= 1e-4, devref   # vs the reference's float32 serial sums (its own error envelope)


def test_em_single_iteration_accumulators(ctx, golden_dir):
    """One EM step with threshold 0 / max_iter 1 against the oracle's fp64 accumulation of the reference's float32
    weights: em_fast = 0 (the reference's three divisions per weight) reproduces the normalised rows to 2 ulp;
    the default em_fast = 1 (one reciprocal per weight) to BASELINE.json's 1e-5 relative."""
    r = cpu_pipeline(golden_dir, "mafk100_w8_both")
    g = r["g"]
    W, K = r["W"], r["K"]
    d = gpu_tables(ctx, r)
    bg_k = pk.DeviceArray.from_host(ctx, d["bgprob"].to_host()[K])
    for fast in (1, 0):
        ctx.set_option("em_fast", fast)
        try:
            pw, iters, _ = ctx.em(W, g["pwm_pre"], d["counts"], bg_k, 1e4, 0.0, 1)
        finally:
            ctx.set_option("em_fast", 1)
        for i in range(len(g["pwm_ids"])):
            acc = po.em_accumulate(W, r["counts"], r["bgp"][K], g["pwm_pre"][i])
            want = acc.astype(np.float32)
            want = want / want.sum(axis=1, keepdims=True, dtype=np.float32)
            assert iters[i] == 1
            if fast:
                assert (np.abs(pw[i].astype(np.float64) - want) <= 1e-5 * np.abs(want)).all()
            else:
                assert ulp_diff(pw[i], want) <= 2


def test_em_many_pwms_batch(ctx, golden_dir):
    """Config-5 shape in miniature: many seed PWMs (0.7 on the seed base, 0.1 elsewhere), threshold 0."""
    r = cpu_pipeline(golden_dir, "mafk100_w8_plus")
    W, K = r["W"], r["K"]
    d = gpu_tables(ctx, r)
    bg_k = pk.DeviceArray.from_host(ctx, d["bgprob"].to_host()[K])
    order = np.lexsort((np.arange(4 ** W), -r["counts"].astype(np.int64)))[:40]
    pwms = np.full((len(order), W, 4), 0.1, np.float32)
    for i, x in enumerate(order):
        for p_ in range(W):
            pwms[i, p_, (int(x) >> (2 * p_)) & 3] = 0.7
    pw, iters, _ = ctx.em(W, pwms, d["counts"], bg_k, 1e4, 0.0, 3)
    assert (iters == 3).all()
    for i in range(0, len(order), 7):
        p1, it1, _ = po.em(W, r["counts"], r["bgp"][K], pwms[i], 1e4, 0.0, 3, mode=1, final_norm=False)
        assert np.abs(pw[i].astype(np.float64) - p1).max() <= 1e-6


@pytest.mark.parametrize("both", [True, False])
def test_iupac_many_large_patterns_in_one_call(ctx, both):
    """N-rich patterns (more members than one workgroup sorts in LDS) take the grouped bitmap / compact / gather / fold
    pipeline, many patterns per round of launches; mixed with small ones, in any order, over several groups (the scratch
    budget is squeezed so that the call needs more than one), every row must equal the oracle's serial sums bit for bit."""
    W = 8
    rng = np.random.default_rng(31 + both)
    codes, offs = po.synth(1, 0, 20000, 120)
    counts, ltot = po.count(codes, offs, W, both)
    V = po.bg_V(po.bg_counts(codes, offs, 2), 2)
    bgp = po.bgprob(W, 2, V, both)
    stats = po.stats(W, counts, bgp, ltot)
    expected = stats[0] if isinstance(stats, tuple) else stats["expected"]
    letters = np.array([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10])
    ids = []
    for _ in range(60):
        n_N = rng.integers(3, 9)  # 3..8 positions are N: 2^6 .. 4^8 members
        pos_N = rng.choice(W, size=n_N, replace=False)
        ls = rng.choice(letters, size=W)
        ls[pos_N] = 10
        ids.append(sum(int(l) * 11 ** i for i, l in enumerate(ls)))
    ids.append(sum(10 * 11 ** i for i in range(W)))  # all N
    ids = np.array(ids, dtype=np.uint64)
    d_counts = pk.DeviceArray.from_host(ctx, counts.astype(np.uint32))
    d_bgp = pk.DeviceArray.from_host(ctx, bgp.astype(np.float32))
    d_exp = pk.DeviceArray.from_host(ctx, np.asarray(expected, np.float32))
    want = [po.iupac_aggregate(int(i), W, both, counts, bgp, expected) for i in ids]
    for budget in (0, 3 << 20):
        ctx.set_option("iupac_group_bytes", budget)
        out = ctx.iupac_aggregate(W, both, ids, d_counts, d_bgp, d_exp)
        for j, w in enumerate(want):
            assert int(out["sites"][j]) == w.sites, (budget, j)
            for f in ("bg_p", "expected", "zscore", "log_pvalue"):
                a = np.float32(out[f][j]).view(np.uint32)
                b = np.float32(getattr(w, f)).view(np.uint32)
                assert a == b, (budget, j, f, po.iupac_str(int(ids[j]), W))
    ctx.set_option("iupac_group_bytes", 0)


def test_em_fast_mode_within_stated_tolerance(ctx, golden_dir):
    """Option em_fast = 1 (the default) replaces the three IEEE divisions of a k-mer weight by one reciprocal.  BASELINE.json's bar for EM
    results is 1e-5 relative: the PWMs after 10 iterations must agree with the exact mode (and so with the oracle) to that,
    with the same iteration counts on the golden cases."""
    for name in ("mafk_w10_both", "mafk100_w8_plus"):
        r = cpu_pipeline(golden_dir, name)
        g = r["g"]
        W, K = r["W"], r["K"]
        if len(g["pwm_ids"]) == 0:
            continue
        d = gpu_tables(ctx, r)
        bg_k = pk.DeviceArray.from_host(ctx, d["bgprob"].to_host()[K])
        fast, it1, ch1 = ctx.em(W, g["pwm_pre"], d["counts"], bg_k, 1e4, 0.08, 10)
        ctx.set_option("em_fast", 0)
        try:
            exact, it0, ch0 = ctx.em(W, g["pwm_pre"], d["counts"], bg_k, 1e4, 0.08, 10)
        finally:
            ctx.set_option("em_fast", 1)
        assert np.array_equal(it0, it1)
        rel = np.abs(fast.astype(np.float64) - exact) / np.maximum(np.abs(exact), 1e-30)
        assert rel.max() <= 1e-5, rel.max()
        assert np.abs(ch1 - ch0).max() <= 1e-5


@pytest.mark.parametrize("n_pwm", [160, 513])
def test_em_large_batch_geometry_against_oracle(ctx, golden_dir, n_pwm):
    """160 and 513 PWMs at W = 10 select the 256-leaves-per-thread geometry (the one BASELINE configs[4] runs on); both weight
    modes against the oracle's fp64-accumulating EM, 2 iterations, for a sample of the PWMs (1e-5 relative, the stated bar)."""
    r = cpu_pipeline(golden_dir, "mafk_w10_plus")
    W, K = r["W"], r["K"]
    d = gpu_tables(ctx, r)
    bg_k = pk.DeviceArray.from_host(ctx, d["bgprob"].to_host()[K])
    order = np.lexsort((np.arange(4 ** W), -r["counts"].astype(np.int64)))[:n_pwm]
    pwms = np.full((len(order), W, 4), 0.1, np.float32)
    for i, x in enumerate(order):
        for p_ in range(W):
            pwms[i, p_, (int(x) >> (2 * p_)) & 3] = 0.7
    want = {i: po.em(W, r["counts"], r["bgp"][K], pwms[i], 1e4, 0.0, 2, mode=1, final_norm=False)[0] for i in (0, 57, n_pwm - 2, n_pwm - 1)}
    for fast in (1, 0):
        ctx.set_option("em_fast", fast)
        try:
            pw, iters, _ = ctx.em(W, pwms, d["counts"], bg_k, 1e4, 0.0, 2)
        finally:
            ctx.set_option("em_fast", 1)
        assert (iters == 2).all()
        for i, p1 in want.items():
            assert (np.abs(pw[i].astype(np.float64) - p1) <= 1e-5 * np.abs(p1)).all(), (fast, i)


@pytest.mark.parametrize("W,both", [(2, True), (2, False), (4, True), (4, False), (12, True), (12, False)])
def test_all_kernels_at_the_ends_of_the_length_range(ctx, W, both):
    """The golden cases cover W = 6, 8, 10.  W = 2 (16 patterns, background order 1, the one-wave EM kernel, with invalid
    bases in the input -- the goldens at W = 2 have few), W = 4 (tables smaller than a workgroup, K = 2 < W - 1 barely) and W = 12
    (two-level count, 2^24 patterns) go through every kernel here against the oracle on a synthetic set: count + sweep
    bit-exact (log-p 1 ulp), IUPAC sums bit-exact, one EM iteration within 1e-5 relative."""
    K = min(W - 1, 2)
    codes, offs = po.synth(5, 0, 3000, 150)
    codes = codes.copy()
    codes[::997] = 0  # some invalid bases: runs, the N rule
    n = po.bg_counts(codes, offs, 2)
    V = po.bg_V(n, 2)
    counts, ltot = po.count(codes, offs, W, both)
    bgp = [po.bgprob(W, k, V, both) for k in range(K + 1)]
    e, lp, z = po.stats(W, counts, bgp[K], ltot)
    r = dict(W=W, both=both, K=K, codes=codes, offs=offs)
    d = gpu_tables(ctx, r)
    assert int(d["ltot"].to_host()[0]) == ltot
    assert bits_equal(d["counts"].to_host().astype(np.uint64), counts)
    got_bgp = d["bgprob"].to_host()
    for k in range(K + 1):
        assert bits_equal(got_bgp[k], bgp[k]), "bgp%d" % k
    assert bits_equal(d["expected"].to_host(), e)
    assert bits_equal(d["z"].to_host(), z)
    assert ulp_diff(d["logp"].to_host(), lp) <= 1
    # K4: a few degenerate patterns around the most frequent k-mer, small and large
    top = int(np.argmax(counts))
    base = [(top >> (2 * q)) & 3 for q in range(W)]
    ids = []
    for deg in ([0], [1, 2], list(range(W // 2)), list(range(W - 1)), list(range(W))):
        ls = list(base)
        for q in deg:
            if q < W:  # (W = 2 has positions 0 and 1 only)
                ls[q] = 10 if q % 2 == 0 else 4 + (q % 6)
        ids.append(sum(l * 11 ** q for q, l in enumerate(ls)))
    ids = np.array(ids, dtype=np.uint64)
    bgp_k = pk.DeviceArray.from_host(ctx, bgp[K])
    out = ctx.iupac_aggregate(W, both, ids, d["counts"], bgp_k, d["expected"])
    for j, i in enumerate(ids):
        w = po.iupac_aggregate(int(i), W, both, counts, bgp[K], e)
        assert int(out["sites"][j]) == w.sites
        for f in ("bg_p", "expected", "zscore", "log_pvalue"):
            assert np.float32(out[f][j]).view(np.uint32) == np.float32(getattr(w, f)).view(np.uint32), (f, po.iupac_str(int(i), W))
    # K5: one iteration from a seed PWM
    pwm = np.full((1, W, 4), 0.1, np.float32)
    for q in range(W):
        pwm[0, q, base[q]] = 0.7
    for fast in (1, 0):
        ctx.set_option("em_fast", fast)
        try:
            pw, iters, _ = ctx.em(W, pwm, d["counts"], bgp_k, 1e4, 0.0, 1)
        finally:
            ctx.set_option("em_fast", 1)
        acc = po.em_accumulate(W, counts, bgp[K], pwm[0])
        want = acc / acc.sum(axis=1, keepdims=True)
        assert iters[0] == 1
        assert (np.abs(pw[0].astype(np.float64) - want) <= 1e-5 * np.abs(want)).all(), fast


@pytest.mark.parametrize("seed", range(48))
def test_count_fuzz_against_oracle(ctx, seed):
    """Random small inputs over the whole parameter space of K1 (W, strand mode, emitter, item length, ragged and
    too-short sequences, invalid bases, low-complexity stretches that trigger the non-overlap rule and the deferred
    fix-up): counts, ltot and -- where the input is made of whole runs -- the fused background counters, bit for bit."""
    rng = np.random.default_rng(1000 + seed)
    # W = 14 (2 GiB oracle tables per case) has its own low-complexity test; PENGK_FUZZ_WS=2,4,... redraws W from another list
    # (tests/tools/count_fuzz.py soaks W = 2 that way: the suite's 48 seeds keep the cases they always had)
    ws = [int(x) for x in os.environ["PENGK_FUZZ_WS"].split(",")] if os.environ.get("PENGK_FUZZ_WS") else [4, 6, 8, 10, 12]
    W = int(rng.choice(ws))
    both = bool(rng.integers(0, 2))
    M = int(rng.choice([0, 64, 100, 256]))
    impl = int(rng.choice([0, 1, 2])) if W in (8, 10, 12) else int(rng.choice([0, 1]))
    n_seq = int(rng.integers(1, 400))
    p_invalid = float(rng.choice([0.0, 0.0, 1e-3, 2e-2]))
    seqs = []
    for _ in range(n_seq):
        kind = rng.integers(0, 10)
        L = int(rng.integers(1, W)) if kind == 0 else int(rng.integers(W, 60)) if kind < 3 else int(rng.integers(60, 700)) \
            if kind < 9 else int(rng.integers(2000, 6000))
        s = rng.integers(1, 5, size=L).astype(np.uint8)
        if rng.random() < 0.3 and L > 40:  # a low-complexity stretch: unit of length 1..12 repeated
            unit = rng.integers(1, 5, size=int(rng.integers(1, 13))).astype(np.uint8)
            a = int(rng.integers(0, L - 20))
            b = min(L, a + int(rng.integers(20, 400)))
            s[a:b] = np.tile(unit, (b - a) // len(unit) + 1)[:b - a]
        if p_invalid:
            s[rng.random(L) < p_invalid] = 0
        seqs.append(s)
    codes = np.concatenate(seqs)
    offs = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.int64)
    want, ltot = po.count(codes, offs, W, both)
    p = pk.Packed(codes, offs, W, M)
    ctx.upload(p)
    ctx.set_option("count_impl", impl)
    try:
        if p.all_whole:
            counts, lt, bg = ctx.count_bg(both)
            assert np.array_equal(bg.to_host().astype(np.int64), po.bg_counts(codes, offs, 2)), "bg"
        else:
            counts, lt = ctx.count(both)
    finally:
        ctx.set_option("count_impl", 0)
    if both:
        ctx.mirror(W, counts)
    assert int(lt.to_host()[0]) == ltot
    got = counts.to_host().astype(np.uint64)
    assert np.array_equal(got, want), "W=%d both=%s M=%d impl=%d: %d bins differ" % (W, both, M, impl, int((got != want).sum()))
    assert np.array_equal(p.bg_counts, po.bg_counts(codes, offs, 2))


@pytest.mark.parametrize("name", ["mafk100_w8_both", "mafk100_w8_plus", "torture_w6_plus", "mafk_w10_both", "mafk_w10_plus"])
def test_em_serial_mode_is_bit_exact(ctx, golden_dir, name):
    """em_fast = 2 reproduces the reference's float32 arithmetic including the ORDER in which it adds the 4^W weights
    of a PWM cell: PWMs, iteration counts and the last `change` equal the oracle's serial mode bit for bit, and after the
    constructor's extra normalisation (src/iupac_pattern.cpp:61) the PWMs the compiled reference produced."""
    r = cpu_pipeline(golden_dir, name)
    g = r["g"]
    W, K = r["W"], r["K"]
    if len(g["pwm_ids"]) == 0:
        pytest.skip("no PWMs in this case")
    d = gpu_tables(ctx, r)
    bg_k = pk.DeviceArray.from_host(ctx, d["bgprob"].to_host()[K])
    ctx.set_option("em_fast", 2)
    try:
        pw, iters, change = ctx.em(W, g["pwm_pre"], d["counts"], bg_k, 1e4, 0.08, 10)
    finally:
        ctx.set_option("em_fast", 1)
    for i in range(len(g["pwm_ids"])):
        p0, it0, ch0 = po.em(W, r["counts"], r["bgp"][K], g["pwm_pre"][i], 1e4, 0.08, 10, mode=0, final_norm=False)
        assert iters[i] == it0 == int(g["em_iters"][i])
        assert bits_equal(pw[i], p0), i
        assert np.float32(change[i]).view(np.uint32) == np.float32(ch0).view(np.uint32)
        fin = pw[i] / pw[i].sum(axis=1, keepdims=True, dtype=np.float32)
        assert bits_equal(fin.astype(np.float32), g["pwm_post"][i].astype(np.float32)), i


def test_em_serial_mode_many_pwms_in_batches(ctx, golden_dir):
    """300 PWMs at W = 10 exceed one batch of the serial mode's weight tables (1 GiB = 128 PWMs with the scan's second,
    permuted copy of each table, 256 without): the later batches must be
    as bit-exact as the first (sampled PWMs against the oracle's serial mode, 2 iterations)."""
    r = cpu_pipeline(golden_dir, "mafk_w10_plus")
    W, K = r["W"], r["K"]
    d = gpu_tables(ctx, r)
    bg_k = pk.DeviceArray.from_host(ctx, d["bgprob"].to_host()[K])
    order = np.lexsort((np.arange(4 ** W), -r["counts"].astype(np.int64)))[:300]
    pwms = np.full((len(order), W, 4), 0.1, np.float32)
    for i, x in enumerate(order):
        for p_ in range(W):
            pwms[i, p_, (int(x) >> (2 * p_)) & 3] = 0.7
    ctx.set_option("em_fast", 2)
    try:
        pw, iters, change = ctx.em(W, pwms, d["counts"], bg_k, 1e4, 0.0, 2)
    finally:
        ctx.set_option("em_fast", 1)
    assert (iters == 2).all()
    for i in (0, 1, 2, 127, 128, 255, 256, 298, 299):
        p0, it0, ch0 = po.em(W, r["counts"], r["bgp"][K], pwms[i], 1e4, 0.0, 2, mode=0, final_norm=False)
        assert bits_equal(pw[i], p0), i
        assert np.float32(change[i]).view(np.uint32) == np.float32(ch0).view(np.uint32)


@pytest.mark.parametrize("W", [12, 14])
def test_em_serial_scan_equals_the_fold_on_large_tables(ctx, W):
    """W = 12 and 14: position 0's cells read every fourth float straight from the weight table (no permuted copy,
    csrc/em.hip EmTerms0) and a cell's chain is 1024 / 16384 blocks long: the scan against the dependent-addition fold,
    bit for bit, on device-generated tables (the oracle's 4^14 tables would take minutes)."""
    ctx.synth(9, 0, 30000, 200, W)
    counts, ltot, bg = ctx.count_bg(True)
    ctx.mirror(W, counts)
    V = ctx.bg_model(bg, 2)
    bgprob, expected, logp, z = ctx.pattern_stats(W, True, 2, 2, V, ltot, counts)
    bg_k = pk.DeviceArray.from_host(ctx, bgprob.to_host()[2])
    rng = np.random.default_rng(W)
    pwms = rng.dirichlet(np.ones(4) * 3, size=(3, W)).astype(np.float32)
    out = {}
    ctx.set_option("em_fast", 2)
    try:
        for scan in (3, 2, 1, 0):
            ctx.test_em_generation(scan)
            out[scan] = ctx.em(W, pwms.copy(), counts, bg_k, 1e4, 0.0, 2)
    finally:
        ctx.test_em_generation(2)
        ctx.set_option("em_fast", 1)
    for scan in (1, 2, 3):  # 2 = the default: blocks evaluated ahead of their chain; 3 = weights and evaluation as one kernel (W = 14: as 2)
        assert out[scan][0].tobytes() == out[0][0].tobytes()
        assert out[scan][1].tolist() == out[0][1].tolist() and out[scan][2].tobytes() == out[0][2].tobytes()
    assert np.isfinite(out[1][0]).all() and not np.array_equal(out[1][0], pwms)


@pytest.mark.parametrize("scheme", [2, 3])
@pytest.mark.parametrize("W", [10, 12])
@pytest.mark.parametrize("skew", [1, 2, 5])
def test_em_serial_blocks_ahead_with_wrong_binade_estimates(ctx, W, skew, scheme):
    """The serial mode evaluates a cell's blocks ahead of the chain under the binade an ESTIMATE of the sum predicts
    (csrc/seqsum.h); exactness must never rest on it.  Test hook em_test_skew = n: about every n-th block is handed
    the binade above the right one, or a binade where none is to be had (n = 1: every block) -- the chain's checks must
    turn every one of them down and fetch the block itself.  Same bits as the dependent-addition fold, on heavy-tailed
    counts (sums that cross many binades) and with 20 PWMs on two streams.  scheme 3 = the two-launch variant ("em_serial_scan"
    = 3: csrc/em.hip, em_span_fused_kernel), whose estimates come from a bounded look-back inside the kernel: there,
    em_test_lookback = 3 * skew also makes workgroups act as if that look-back had timed out."""
    NP = 4 ** W
    rng = np.random.default_rng(100 * W + skew)
    c = rng.lognormal(1.0, 2.5, NP).astype(np.uint32)
    bg = (rng.dirichlet(np.ones(64), size=NP // 64).reshape(-1)).astype(np.float32)
    counts, bgd = pk.DeviceArray.from_host(ctx, c), pk.DeviceArray.from_host(ctx, bg)
    pwms = np.maximum(rng.dirichlet(np.full(4, 0.5), size=(20 if W == 10 else 4, W)).astype(np.float32), np.float32(1e-20))
    ctx.set_option("em_fast", 2)
    try:
        ctx.test_em_generation(0)
        ref = ctx.em(W, pwms, counts, bgd, 1e4, 0.0, 3)
        ctx.test_em_generation(scheme)
        ctx.set_option("em_test_skew", skew)
        ctx.set_option("em_test_lookback", 3 * skew if scheme == 3 else 0)
        got = ctx.em(W, pwms, counts, bgd, 1e4, 0.0, 3)
        met = {k: ctx.info("em_" + k) for k in ("fetched_blocks", "mispredicted_blocks", "restaged_blocks", "restaged_waits")}
    finally:
        ctx.set_option("em_test_skew", 0)
        ctx.set_option("em_test_lookback", 0)
        ctx.test_em_generation(2)
        ctx.set_option("em_fast", 1)
    assert got[0].tobytes() == ref[0].tobytes()
    assert got[1].tolist() == ref[1].tolist() and got[2].tobytes() == ref[2].tobytes()
    assert np.isfinite(got[0]).all()
    # ... and the branches in question did run: blocks whose binade did not hold were fetched on demand, some of them
    # through the buffer of a block already asked for (which is then asked for again: the walk's `restaged` state)
    cells = len(pwms) * 4 * W * 3
    assert met["mispredicted_blocks"] > cells and met["fetched_blocks"] > met["mispredicted_blocks"], met
    assert met["restaged_blocks"] > 0, met
    if W == 12:
        assert met["restaged_waits"] > 0, met


@pytest.mark.parametrize("W", [10, 12, 14])
def test_em_head_blocks_folded_beside_the_evaluation(ctx, W):
    """Option em_head_blocks = K: the first K blocks of every cell are folded one after the other from zero by extra
    workgroups of em_span_eval_kernel (record 0 carries the sum and K, the chain starts behind them; csrc/seqsum.h
    walk_chain).  K = 1 (the default), 2, 4, 7 and 64 (a whole 64-block chunk: at W = 10 the whole cell)
    give the dependent-addition fold's bits, also with wrong binade estimates; fewer blocks are left to the chains."""
    NP = 4 ** W
    rng = np.random.default_rng(7 * W)
    if W < 14:
        c = rng.lognormal(1.0, 2.5, NP).astype(np.uint32)
        bg = (rng.dirichlet(np.ones(64), size=NP // 64).reshape(-1)).astype(np.float32)
        counts, bgd = pk.DeviceArray.from_host(ctx, c), pk.DeviceArray.from_host(ctx, bg)
    else:  # (device-generated tables: 4^14 entries)
        ctx.synth(11, 0, 30000, 200, W)
        counts, ltot, bgc = ctx.count_bg(True)
        ctx.mirror(W, counts)
        bgprob = ctx.pattern_stats(W, True, 2, 2, ctx.bg_model(bgc, 2), ltot, counts)[0]
        bgd = pk.DeviceArray.from_host(ctx, bgprob.to_host()[2])
    pwms = np.maximum(rng.dirichlet(np.full(4, 0.5), size=(10 if W == 10 else 3 if W == 12 else 1, W)).astype(np.float32), np.float32(1e-20))
    default = 1  # (pengk_internal.h: em_head_blocks)
    ctx.set_option("em_fast", 2)
    fetched = {}
    try:
        ctx.test_em_generation(0 if W < 14 else 1)
        ref = ctx.em(W, pwms, counts, bgd, 1e4, 0.0, 2)
        ctx.test_em_generation(2)
        for K, skew in ((1, 0), (2, 0), (4, 0), (7, 3), (64, 0)) if W < 14 else ((1, 0), (64, 3)):
            ctx.set_option("em_head_blocks", K)
            ctx.set_option("em_test_skew", skew)
            got = ctx.em(W, pwms, counts, bgd, 1e4, 0.0, 2)
            fetched[K] = ctx.info("em_fetched_blocks")
            assert got[0].tobytes() == ref[0].tobytes(), K
            assert got[1].tolist() == ref[1].tolist() and got[2].tobytes() == ref[2].tobytes(), K
        for bad in (0, 65):
            with pytest.raises(Exception):
                ctx.set_option("em_head_blocks", bad)
    finally:
        ctx.set_option("em_test_skew", 0)
        ctx.set_option("em_head_blocks", default)
        ctx.set_option("em_fast", 1)
    if W < 14:
        assert fetched[1] > fetched[2] > fetched[4], fetched
    if W == 10:
        assert fetched[64] == 0, fetched  # (64 blocks per cell: nothing left for the chains)


@pytest.mark.parametrize("W,n", [(10, 1), (10, 9), (10, 40), (12, 9)])
def test_em_serial_iteration_protocol_at_its_edges(ctx, W, n):
    """The blocks-ahead scheme keeps the finalize step of iteration k at the head of iteration k + 1's first kernel, flags
    and PWMs double-buffered by iteration parity, the last one in em_serial_finish_kernel (csrc/em.hip).  Its edges: 0, 1, 2
    and 11 iterations; thresholds that stop nobody, some PWMs early, PWMs after one iteration, everybody before the first; one PWM, a partial
    group of eight, several batches (table budget 20 MiB = 5 PWMs); and the same call twice on one context (no flag may
    survive a call).  Against the dependent-addition fold: PWMs, iteration counts and `change` bit for bit."""
    NP = 4 ** W
    rng = np.random.default_rng(1000 * W + n)
    c = rng.poisson(2.0, NP).astype(np.uint32)
    bg = (rng.dirichlet(np.ones(64), size=NP // 64).reshape(-1)).astype(np.float32)
    counts, bgd = pk.DeviceArray.from_host(ctx, c), pk.DeviceArray.from_host(ctx, bg)
    pw = rng.dirichlet(np.ones(4) * 2, size=(n, W)).astype(np.float32)
    pw[0] = np.float32(0.25)
    seen = set()
    ctx.set_option("em_fast", 2)
    try:
        for max_it in ((-5, 0, 1, 2, 11) if W == 10 else (0, 1, 3)):  # (negative: the reference's loop never runs, like 0)
            for thr in (0.0, 0.05, W - 0.5, 1e9):
                for budget in ((0, 20) if n == 40 else (0,)):
                    ctx.set_option("em_table_budget_mb", budget)
                    ctx.test_em_generation(0)
                    ref = ctx.em(W, pw, counts, bgd, 1e4, thr, max_it)
                    ctx.test_em_generation(2)
                    for again in range(2):
                        got = ctx.em(W, pw, counts, bgd, 1e4, thr, max_it)
                        assert got[1].tolist() == ref[1].tolist(), (max_it, thr, budget, again)
                        assert got[0].tobytes() == ref[0].tobytes() and got[2].tobytes() == ref[2].tobytes(), (max_it, thr, budget, again)
                    seen.update(ref[1].tolist())
                    if max_it <= 0:
                        assert ref[0].tobytes() == pw.tobytes() and set(ref[1].tolist()) == {0}
                    if thr == 1e9:  # (the loop starts from change = W, src/peng.cpp:101-106: nothing runs)
                        assert ref[0].tobytes() == pw.tobytes() and set(ref[1].tolist()) == {0}
                    if thr == W - 0.5 and max_it > 0:
                        assert min(ref[1].tolist()) == 1
    finally:
        ctx.set_option("em_table_budget_mb", 0)
        ctx.test_em_generation(2)
        ctx.set_option("em_fast", 1)
    if n >= 9 and W == 10:
        assert len(seen) >= 5, seen  # PWMs leave at different iterations


def _ramp_counts(W, doublings=30, blocks_per_doubling=16):
    """A count table whose running sums cross a power of two about every `blocks_per_doubling` blocks of EVERY cell's
    chain (a block = 4096 terms of a cell = 16384 consecutive x), `doublings` times, then stay flat: several blocks
    without a binade in most 64-block chunks of the first part of a chain -- what the chain walk's double buffering and
    its re-requests are there for, and what real count tables produce once in a few thousand PWMs."""
    x = np.arange(4 ** W, dtype=np.float64)
    period = blocks_per_doubling * 16384.0
    return np.floor(2.0 ** np.minimum(x / period, float(doublings))).astype(np.uint32)


@pytest.mark.parametrize("kind,skew", [("lognormal", 0), ("lognormal", 3), ("ramp", 0)])
def test_em_blocks_ahead_at_w12_against_the_oracle_on_tables_that_cross_many_binades(ctx, kind, skew):
    """The blocks-ahead EM at W = 12 (1024 blocks per cell, 16 chunks of block records per chain) against the ORACLE's
    left-to-right float32 sums, bit for bit -- not against this library's own dependent-addition fold -- on tables whose
    sums cross many binades: heavy-tailed lognormal counts, and a ramp that doubles the running sum every 16 blocks.
    (Reference: /root/reference/src/peng.cpp:104-144.)"""
    W = 12
    NP = 4 ** W
    rng = np.random.default_rng(1200 + skew)
    c = rng.lognormal(1.0, 2.5, NP).astype(np.uint32) if kind == "lognormal" else _ramp_counts(W)
    bg = (rng.dirichlet(np.ones(64), size=NP // 64).reshape(-1)).astype(np.float32)
    pwms = np.maximum(rng.dirichlet(np.full(4, 0.5 if kind == "lognormal" else 40.0), size=(3, W)).astype(np.float32), np.float32(1e-20))
    counts, bgd = pk.DeviceArray.from_host(ctx, c), pk.DeviceArray.from_host(ctx, bg)
    ctx.set_option("em_fast", 2)
    try:
        ctx.set_option("em_test_skew", skew)
        got, iters, change = ctx.em(W, pwms, counts, bgd, 1e4, 0.0, 2)
        met = {k: ctx.info("em_" + k) for k in ("fetched_blocks", "mispredicted_blocks", "restaged_blocks", "restaged_waits")}
    finally:
        ctx.set_option("em_test_skew", 0)
        ctx.set_option("em_fast", 1)
    c64 = c.astype(np.uint64)
    for i in range(len(pwms)):
        ref, it, ch = po.em(W, c64, bg, pwms[i], 1e4, 0.0, 2, mode=0, final_norm=False)
        assert iters[i] == it == 2
        assert got[i].tobytes() == ref.astype(np.float32).tobytes(), (kind, skew, i)
        assert np.float32(change[i]).view(np.uint32) == np.float32(ch).view(np.uint32)
    chains = len(pwms) * 4 * W * 2
    # (block 0 of a chain is folded from zero beside the block evaluation and not counted; on these tables every chain
    # still meets blocks behind it in which its sum passes a power of two)
    assert met["fetched_blocks"] >= chains, met
    if kind == "ramp":
        assert met["fetched_blocks"] > 10 * chains, met  # ~30 crossings per chain
    if skew:
        assert met["mispredicted_blocks"] > 0 and met["restaged_blocks"] > 0, met


def test_em_chain_walk_restaged_block_is_pinned(ctx):
    """The defect round 3's soak found once in ~2000 random W = 12 cases, walked on purpose (csrc/seqsum.h, walk_chain):
    a chunk of 64 block records holds blocks A < B without a binade -- both asked for when the chunk starts, A into
    buffer a, B into buffer b -- and, in FRONT of A, a block M whose binade does not hold.  M is fetched on demand
    through buffer a; A is asked for again and its loads are now YOUNGER than B's, so the wait in front of A's rows must
    be vmcnt(0): the counted vmcnt(16), right for the usual order, covers B only, and the wave would read M's rows as
    A's.  The ramp table puts ~4 blocks without a binade into most chunks of a chain's first half, em_test_skew = 3
    makes every third block a wrong guess: `restaged_waits` counts the takes of a re-requested block with another block
    staged behind it, and the PWMs must be the oracle's bit for bit.  (A build with -DPENGK_TEST_WALK_RACE puts the
    counted wait back: this test fails on it, profiles/r04_walk_race_pin.log.)"""
    W = 12
    c = _ramp_counts(W)
    rng = np.random.default_rng(77)
    bg = np.full(4 ** W, np.float32(1.0 / 4 ** W))
    pwms = rng.dirichlet(np.full(4, 60.0), size=(4, W)).astype(np.float32)
    counts, bgd = pk.DeviceArray.from_host(ctx, c), pk.DeviceArray.from_host(ctx, bg)
    ctx.set_option("em_fast", 2)
    try:
        ctx.set_option("em_test_skew", 3)
        got, iters, change = ctx.em(W, pwms, counts, bgd, 1e4, 0.0, 2)
        met = {k: ctx.info("em_" + k) for k in ("fetched_blocks", "mispredicted_blocks", "restaged_blocks", "restaged_waits")}
    finally:
        ctx.set_option("em_test_skew", 0)
        ctx.set_option("em_fast", 1)
    assert met["restaged_waits"] >= 4 * 4 * W, met  # at least once per chain on average (first iteration alone)
    c64 = c.astype(np.uint64)
    for i in range(len(pwms)):
        ref, it, ch = po.em(W, c64, bg, pwms[i], 1e4, 0.0, 2, mode=0, final_norm=False)
        assert iters[i] == it == 2
        assert got[i].tobytes() == ref.astype(np.float32).tobytes(), i
        assert np.float32(change[i]).view(np.uint32) == np.float32(ch).view(np.uint32)


def test_em_serial_batches_on_several_streams(ctx, golden_dir):
    """Option em_overlap: the serial mode's batches of PWMs take turns on 1..4 streams (csrc/em.hip, launch_serial_ahead).
    41 PWMs at W = 10 with a table budget of 32 MiB = batches of 8 / 4 / 2 PWMs: every stream count gives the bits of the
    dependent-addition fold, also with PWMs that stop early (a threshold half of them meet by the third iteration) and a PWM count that leaves a ragged
    last batch."""
    r = cpu_pipeline(golden_dir, "mafk_w10_both")
    W, K = r["W"], r["K"]
    d = gpu_tables(ctx, r)
    bg_k = pk.DeviceArray.from_host(ctx, d["bgprob"].to_host()[K].copy())
    rng = np.random.default_rng(41)
    pwms = rng.dirichlet(np.ones(4) * 2, size=(41, W)).astype(np.float32)
    order = np.lexsort((np.arange(4 ** W), -r["counts"].astype(np.int64)))[:10]
    for i, x in enumerate(order):   # ten seed-like PWMs: these converge within a few iterations
        for p_ in range(W):
            pwms[i, p_] = 0.1
            pwms[i, p_, (int(x) >> (2 * p_)) & 3] = 0.7
    out = {}
    ctx.set_option("em_fast", 2)
    ctx.set_option("em_table_budget_mb", 32)
    try:
        ctx.test_em_generation(0)
        thr = float(np.median(ctx.em(W, pwms, d["counts"], bg_k, 1e4, 0.0, 3)[2]))  # half of the PWMs stop after <= 3 iterations
        ref = ctx.em(W, pwms, d["counts"], bg_k, 1e4, thr, 6)
        for scheme in (2, 3):
            ctx.test_em_generation(scheme)
            for streams in (1, 2, 3, 4):
                ctx.set_option("em_overlap", streams)
                out[(scheme, streams)] = ctx.em(W, pwms, d["counts"], bg_k, 1e4, thr, 6)
    finally:
        ctx.test_em_generation(2)
        ctx.set_option("em_overlap", 2)
        ctx.set_option("em_table_budget_mb", 0)
        ctx.set_option("em_fast", 1)
    assert len(set(ref[1].tolist())) > 1   # not all PWMs ran the same number of iterations
    for streams, got in out.items():
        assert got[0].tobytes() == ref[0].tobytes(), streams
        assert got[1].tolist() == ref[1].tolist() and got[2].tobytes() == ref[2].tobytes(), streams


@pytest.mark.parametrize("name", ["mafk100_w8_both", "mafk_w10_plus"])
def test_em_serial_scan_equals_the_dependent_addition_fold(ctx, golden_dir, name):
    """The serial mode sums a cell's weights with a wave-wide scan (csrc/seqsum.h; generation 1 of pengk_test_em_generation, the product's path at W = 8)
    or, as in the first rounds, by dependent additions (0).  Both must give the same bits on every PWM -- including
    degenerate PWMs whose weights are NaN (a zero PWM entry over a zero background entry: 0 / 0), which the scan hands
    to the dependent-addition fold PWM by PWM."""
    r = cpu_pipeline(golden_dir, name)
    W, K = r["W"], r["K"]
    d = gpu_tables(ctx, r)
    bgk = d["bgprob"].to_host()[K].copy()
    order = np.lexsort((np.arange(4 ** W), -r["counts"].astype(np.int64)))[:40]
    pwms = np.full((len(order), W, 4), 0.1, np.float32)
    for i, x in enumerate(order):
        for p_ in range(W):
            pwms[i, p_, (int(x) >> (2 * p_)) & 3] = 0.7
    rng = np.random.default_rng(4)
    pwms[5:20] = rng.dirichlet(np.ones(4), size=(15, W)).astype(np.float32)
    pwms[7, 3] = [0.0, 0.5, 0.5, 0.0]     # zero PWM entries: weights of 0 for three quarters... and NaN where bg is 0 too
    pwms[8, 0] = [1.0, 0.0, 0.0, 0.0]
    bgk[5] = 0.0                          # k-mer 5 = "CCAAAA..." : digit 0 is C, digit 1 is C
    bgk[12345 % (4 ** W)] = 0.0
    bg_k = pk.DeviceArray.from_host(ctx, bgk)
    out = {}
    ctx.set_option("em_fast", 2)
    try:
        for scan in (2, 1, 0):
            ctx.test_em_generation(scan)
            out[scan] = ctx.em(W, pwms, d["counts"], bg_k, 1e4, 0.0, 3)
    finally:
        ctx.test_em_generation(2)
        ctx.set_option("em_fast", 1)
    for scan in (1, 2):
        assert out[scan][0].tobytes() == out[0][0].tobytes()
        assert out[scan][1].tolist() == out[0][1].tolist() and out[scan][2].tobytes() == out[0][2].tobytes()
    assert np.isnan(out[1][0][8]).any() or np.isnan(out[1][0][7]).any()   # the degenerate ones did go through NaN weights
    assert np.isfinite(out[1][0][0]).all()
    # and a healthy PWM of the same batch still matches the oracle
    p0, it0, ch0 = po.em(W, r["counts"], bgk, pwms[0], 1e4, 0.0, 3, mode=0, final_norm=False)
    assert bits_equal(out[1][0][0], p0)


@pytest.mark.parametrize("name", ["mafk_w10_both", "mafk_w10_plus", "mafk100_w8_both"])
def test_seed_candidates_on_device(ctx, golden_dir, name):
    """pengk_seed_candidates compacts exactly the ids the reference's walk can reach (z >= threshold, count >= threshold,
    src/base_pattern.cpp:463-466); ranked by (z descending, id ascending) and walked with the reference's seen /
    neighbour logic they give the golden seed list up to the strand a reverse-complement pair is named on."""
    import ctypes as C
    r = cpu_pipeline(golden_dir, name)
    W, both = r["W"], r["both"]
    d = gpu_tables(ctx, r)
    zthr, cthr = 10.0, 3
    cap = 1 << 16
    ids = np.zeros(cap, np.uint32)
    zs = np.zeros(cap, np.float32)
    n = C.c_int64()
    pk._check(pk.lib().pengk_seed_candidates(ctx.h, W, pk._ptr(d["z"]), pk._ptr(d["counts"]), zthr, cthr, ids.ctypes.data,
                                             zs.ctypes.data, cap, C.byref(n)))
    want = np.nonzero((r["z"] >= zthr) & (r["counts"] >= cthr))[0]
    got = np.sort(ids[: n.value])
    assert n.value == len(want) and np.array_equal(got, want.astype(np.uint32))
    assert np.array_equal(zs[: n.value], r["z"][ids[: n.value]])
    # a too small buffer reports the full count
    n2 = C.c_int64()
    ids4, zs4 = np.zeros(4, np.uint32), np.zeros(4, np.float32)  # (own buffers: the device hands candidates out in any order)
    pk._check(pk.lib().pengk_seed_candidates(ctx.h, W, pk._ptr(d["z"]), pk._ptr(d["counts"]), zthr, cthr, ids4.ctypes.data,
                                             zs4.ctypes.data, 4, C.byref(n2)))
    assert n2.value == n.value
    if not both:
        # single strand: exact z ties between DIFFERENT k-mers that are Hamming neighbours of each other are broken by
        # the reference's non-stable sort; (z desc, id asc) may keep the other one (the documented difference of the
        # opt-in mode).  The candidate set above is what is exact.
        return
    # the walk over (z desc, id asc): same seeds as the reference's ranking up to strand
    order = np.lexsort((ids[: n.value], -zs[: n.value].astype(np.float64)))
    seen = np.zeros(4 ** W, bool)
    sel = []
    for x in ids[: n.value][order].tolist():
        if seen[x] or (both and seen[po.revcomp(x, W)]):
            continue
        sel.append(x)
        seen[x] = True
        for p_ in range(W):
            for c in range(4):
                seen[(x & ~(3 << (2 * p_))) | (c << (2 * p_))] = True
    canon = lambda x: min(x, po.revcomp(x, W)) if both else x  # noqa: E731
    ref_seeds = po.select(W, r["z"], r["counts"], zthr, cthr, not both, True)
    assert sorted(canon(x) for x in sel) == sorted(canon(int(x)) for x in ref_seeds)


def _exact_S(p1, c1, s1, p2, c2, s2, both, bg):
    """IUPACPattern::calculate_S restated with its float32 running sums (src/iupac_pattern.cpp:538-615)."""
    f32, f64 = np.float32, np.float64
    eps = f32(1e-4)

    def term(x, y):
        mean = f32(f32(f32(x + y) + f32(2) * eps) / f32(2))
        return (f64(f32(x + eps)) * np.log2(f64(f32(x + eps))) + f64(f32(y + eps)) * np.log2(f64(f32(y + eps)))
                - f64(f32(f32(2) * mean)) * np.log2(f64(mean)))

    def d(a, b, oa, ob, n):
        acc = f32(0)
        for i in range(n):
            for k in range(4):
                acc = f32(f64(acc) + term(a[oa + i][k], b[ob + i][k]))
        return acc

    def dbg(a, oa, n):
        acc = f32(0)
        for i in range(n):
            for k in range(4):
                acc = f32(f64(acc) + term(a[oa + i][k], bg[k]))
        return acc

    big, small = (p1, c1, s1), (p2, c2, s2)
    if len(p1) < len(p2):
        big, small = small, big
    lb, ls = len(big[0]), len(small[0])
    best = -np.inf
    for orient in range(2 if both else 1):
        pb, ps = big[0], small[0]
        if orient == 1:
            if big[2] < small[2]:
                pb = big[1]
            else:
                ps = small[1]
        for shift in range(6 - ls, lb - 6 + 1):
            off_s, off_b = -min(shift, 0), max(shift, 0)
            ov = min(lb - off_b, ls - off_s)
            sc = f32(0.5 * f64(f32(dbg(pb, off_b, ov) + dbg(ps, off_s, ov))) - f64(d(pb, ps, off_b, off_s, ov)))
            if sc > best:
                best = sc
    return best


@pytest.mark.parametrize("both", [True, False])
def test_motif_similarity_grid_within_margin_of_the_reference_arithmetic(ctx, both):
    """pengk_motif_similarity (fp64 on the device) against calculate_S restated with the reference's float32 running
    sums, on motifs of mixed lengths (merged motifs are longer than W): every pair within 5e-4 -- a quarter of the
    margin the host mirror uses to pick the pairs it evaluates exactly -- for the whole triangle and for the
    one-new-motif column that follows a merge."""
    rng = np.random.default_rng(11)
    n, ML = 40, 64
    lens = rng.integers(6, 15, size=n).astype(np.int32)
    lens[:8] = 10
    pw = np.zeros((n, ML, 4), np.float32)
    cp = np.zeros((n, ML, 4), np.float32)
    for i in range(n):
        m = rng.dirichlet(np.full(4, 0.3), size=lens[i]).astype(np.float32)
        m = np.maximum(m, np.float32(1e-8))
        pw[i, :lens[i]] = m
        cp[i, :lens[i]] = m[::-1, ::-1]
    sites = rng.integers(10, 5000, size=n).astype(np.uint64)
    sites[3] = sites[4]  # equal site counts: the SMALL motif is complemented
    bg = np.array([0.27, 0.23, 0.21, 0.29], np.float32)
    out = np.zeros(n * (n - 1) // 2, np.float32)
    pk._check(pk.lib().pengk_motif_similarity(ctx.h, n, pw.ctypes.data, cp.ctypes.data, lens.ctypes.data, sites.ctypes.data,
                                              int(both), bg.ctypes.data, 0, out.ctypes.data))
    q, worst = 0, 0.0
    for j in range(n):
        for i in range(j):
            want = _exact_S(pw[i, :lens[i]], cp[i, :lens[i]], sites[i], pw[j, :lens[j]], cp[j, :lens[j]], sites[j], both, bg)
            worst = max(worst, abs(float(out[q]) - float(want)))
            q += 1
    assert worst <= 5e-4, worst
    col = np.zeros(n - 1, np.float32)
    pk._check(pk.lib().pengk_motif_similarity(ctx.h, n, pw.ctypes.data, cp.ctypes.data, lens.ctypes.data, sites.ctypes.data,
                                              int(both), bg.ctypes.data, n - 1, col.ctypes.data))
    assert np.array_equal(col, out[-(n - 1):])


def test_lean_division_is_the_ieee_division_on_its_domain(ctx):
    """The serial EM's weights kernel runs the IEEE division's own instruction sequence without its range scaling where a
    PWM's operand ranges allow (csrc/em.hip, lean_div / lean_ranges_ok; the three divisions of
    /root/reference/src/peng.cpp:124-125,186).  pengk_selftest_division draws ~3e9 random operand pairs -- exponents over the
    whole range the guard admits, random and special mantissas, zero numerators -- and compares with the compiler's
    division bit for bit: no pair may differ."""
    import ctypes as C
    total = bad = zeros = 0
    for seed in (1, 2, 3):
        out = (C.c_uint64 * 3)()
        pk._check(pk.lib().pengk_selftest_division(ctx.h, seed, 1024, out))
        total, bad, zeros = total + out[0], bad + out[1], zeros + out[2]
    assert total > 10 ** 9 and zeros > 10 ** 6 and bad == 0, (total, bad, zeros)


@pytest.mark.parametrize("W", [10, 12])
def test_em_weights_with_and_without_the_lean_division(ctx, W):
    """Same PWMs, same tables, option em_lean_div 1 / 0: the same bits -- on ordinary PWMs (which take the lean path), on
    PWMs with entries of 1e-12 and 1e-30 (products far below the guard: the plain divisions), on a background table with a
    zero entry (the whole call takes the plain divisions), and against the oracle for the first of them."""
    NP = 4 ** W
    rng = np.random.default_rng(31 + W)
    c = rng.poisson(3.0, NP).astype(np.uint32)
    c[rng.integers(0, NP, NP // 3)] = 0
    bg = (rng.dirichlet(np.ones(64), size=NP // 64).reshape(-1)).astype(np.float32)
    pwms = rng.dirichlet(np.ones(4) * 2, size=(6, W)).astype(np.float32)
    pwms[3, :, 0] = np.float32(1e-12)
    pwms[4, 2, 1] = np.float32(1e-30)
    pwms[5] = np.float32(0.25)
    counts = pk.DeviceArray.from_host(ctx, c)
    got = {}
    ctx.set_option("em_fast", 2)
    try:
        for tag, table in (("bg", bg), ("bg0", np.where(np.arange(NP) == 12345, np.float32(0), bg))):
            bgd = pk.DeviceArray.from_host(ctx, table)
            for lean in (1, 0):
                ctx.set_option("em_lean_div", lean)
                got[tag, lean] = ctx.em(W, pwms, counts, bgd, 1e4, 0.0, 2)
    finally:
        ctx.set_option("em_lean_div", 1)
        ctx.set_option("em_fast", 1)
    for tag in ("bg", "bg0"):
        a, b = got[tag, 1], got[tag, 0]
        assert a[0].tobytes() == b[0].tobytes() and a[1].tolist() == b[1].tolist() and a[2].tobytes() == b[2].tobytes(), tag
    ref, it, ch = po.em(W, c.astype(np.uint64), bg, pwms[0], 1e4, 0.0, 2, mode=0, final_norm=False)
    assert got["bg", 1][0][0].tobytes() == ref.astype(np.float32).tobytes()


@pytest.mark.parametrize("name", ["torture_w2_both", "mafk100_w2_plus"])
def test_w2_iupac_and_em_against_the_oracle(ctx, golden_dir, name):
    """W = 2 (the shortest pattern length the reference accepts, /root/reference/src/Global.cpp:103-106): no golden run
    has a seed at W = 2 -- the order-1 background models 2-mers exactly -- so the IUPAC aggregation (all 121 two-letter
    IUPAC patterns) and the EM (the one-wave kernel em_w2_kernel, all three modes) are checked against the oracle on the
    reference-pinned tables of the golden case: aggregates and serial-mode PWMs bit for bit, the fp64 modes within 1e-6."""
    r = cpu_pipeline(golden_dir, name)
    W, K, both = r["W"], r["K"], r["both"]
    assert W == 2
    d = gpu_tables(ctx, r)
    bgp_k = pk.DeviceArray.from_host(ctx, d["bgprob"].to_host()[K])
    ids = np.arange(121, dtype=np.uint64)
    out = ctx.iupac_aggregate(W, both, ids, d["counts"], bgp_k, d["expected"])
    for i in ids.tolist():
        st = po.iupac_aggregate(i, W, both, r["counts"], r["bgp"][K], r["expected"])
        assert out["sites"][i] == st.sites == po.iupac_count(i, W, both, r["counts"])
        got = np.array([out["bg_p"][i], out["expected"][i], out["zscore"][i]], np.float32).view(np.uint32)
        want = np.array([st.bg_p, st.expected, st.zscore], np.float32).view(np.uint32)
        assert np.array_equal(got, want), po.iupac_str(i, W)
    rng = np.random.default_rng(2)
    pwms = rng.dirichlet(np.ones(4), size=(7, W)).astype(np.float32)
    pwms[6] = np.float32(0.25)
    for mode, omode, tol in ((2, 0, 0.0), (0, 1, 1e-6), (1, 1, 1e-5)):
        ctx.set_option("em_fast", mode)
        try:
            got, iters, change = ctx.em(W, pwms, d["counts"], bgp_k, 1e4, 0.01, 10)
        finally:
            ctx.set_option("em_fast", 1)
        for i in range(len(pwms)):
            ref, it, ch = po.em(W, r["counts"], r["bgp"][K], pwms[i], 1e4, 0.01, 10, mode=omode, final_norm=False)
            if tol == 0.0:
                assert iters[i] == it and got[i].tobytes() == ref.astype(np.float32).tobytes(), (mode, i)
                assert np.float32(change[i]).view(np.uint32) == np.float32(ch).view(np.uint32)
            else:
                assert abs(int(iters[i]) - it) <= (0 if mode == 0 else 1) and np.abs(got[i].astype(np.float64) - ref).max() <= tol * 10, (mode, i)
