"""pengk_sequential_sum_f32 / the scan behind the EM's serial mode (peng-motif_amd/csrc/seqsum.h).

The device evaluates s <- fl(s + t_i) for a whole chain with a wave-wide scan; the claim is bit-exactness against a
left-to-right float32 loop (numpy's cumulative sum is that loop).  The inputs below are chosen to hit what the proof in
seqsum.h leans on: ties (round-to-nearest-even decided by the parity of the running sum), binade crossings inside every
part of a block, terms far above and far below the running sum, denormals, overflow, and the fallback for terms the
scan does not take.
"""
import numpy as np
import pytest

import peng_motif_amd as pk

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = pk.Context(0)
    yield c
    c.close()


def _ref(rows):
    rows = np.atleast_2d(np.asarray(rows, np.float32))
    if rows.shape[1] == 0:
        return np.zeros(rows.shape[0], np.float32)
    with np.errstate(over="ignore", invalid="ignore"):
        return np.cumsum(rows, axis=1, dtype=np.float32)[:, -1]


def _same(got, want):
    assert got.dtype == np.float32 and want.dtype == np.float32
    assert got.view(np.uint32).tolist() == want.view(np.uint32).tolist()


def _check(ctx, rows):
    rows = np.atleast_2d(np.asarray(rows, np.float32))
    _same(ctx.sequential_sum(rows), _ref(rows))


def test_numpy_cumsum_is_the_left_to_right_loop():
    rng = np.random.default_rng(5)
    t = (rng.random(5000, dtype=np.float32) * np.float32(3.0)).astype(np.float32)
    s = np.float32(0)
    for v in t:
        s = np.float32(s + v)
    assert _ref(t)[0].view(np.uint32) == s.view(np.uint32)


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 4095, 4096, 4097, 8192, 20000, 262144])
def test_uniform_terms_every_length(ctx, n):
    rng = np.random.default_rng(n + 1)
    _check(ctx, rng.random((7, n), dtype=np.float32))


def test_terms_over_the_whole_exponent_range(ctx):
    rng = np.random.default_rng(11)
    for lo, hi in ((-149, -100), (-140, 0), (-30, 30), (-60, 100), (0, 127), (-149, 127)):
        e = rng.integers(lo, hi, size=(16, 30000))
        m = 1.0 + rng.random((16, 30000))
        with np.errstate(over="ignore", under="ignore"):
            t = np.ldexp(m, e).astype(np.float32)
        t[~np.isfinite(t)] = np.float32(0)
        _check(ctx, t)


def test_ties_and_parity(ctx):
    rng = np.random.default_rng(3)
    rows = []
    # a large head, then exact half-ulps: every addition is a tie, decided by the parity of the running sum
    for head in (1.0, 1.0 + 2.0 ** -23, 3.0, 1.5, 2.0 ** 20):
        ulp = np.spacing(np.float32(head))
        t = np.full(9000, ulp / 2, np.float32)
        t[0] = head
        rows.append(t)
        t2 = t.copy()
        t2[1:] = rng.choice(np.array([ulp / 2, ulp, 1.5 * ulp, ulp / 4, 0.0], np.float32), 8999)
        rows.append(t2)
    _check(ctx, np.array(rows))
    # small integers scaled by a power of two: sums stay exactly representable for a while, then start rounding
    k = rng.integers(0, 4, size=(12, 70000)).astype(np.float32)
    _check(ctx, k * np.float32(2.0 ** -12))
    _check(ctx, k)
    # powers of two only
    e = rng.integers(-20, 4, size=(12, 50000))
    _check(ctx, np.ldexp(1.0, e).astype(np.float32))


def test_sparse_chains_and_late_giants(ctx):
    rng = np.random.default_rng(8)
    t = np.zeros((10, 100000), np.float32)
    _check(ctx, t)
    for r in range(10):
        idx = rng.integers(0, t.shape[1], 50 * (r + 1))
        t[r, idx] = rng.random(idx.size, dtype=np.float32) * np.float32(10.0 ** rng.integers(-6, 6))
    _check(ctx, t)
    u = rng.random((6, 40000), dtype=np.float32)
    for r in range(6):
        u[r, rng.integers(0, 40000, 12)] = np.float32(10.0 ** (r + 3))  # single terms far above the running sum
    _check(ctx, u)
    # a crossing in every lane's stretch of the first blocks: the sum doubles again and again
    g = np.ldexp(1.0, np.arange(0, 120) // 1).astype(np.float32)
    _check(ctx, np.concatenate([g, rng.random(9000, dtype=np.float32)]))
    _check(ctx, np.repeat(g, 64))
    _check(ctx, np.repeat(g, 37))


def test_denormals_and_overflow(ctx):
    rng = np.random.default_rng(21)
    tiny = np.float32(1e-45)
    d = (rng.integers(0, 5, size=(8, 30000)).astype(np.float32) * tiny).astype(np.float32)
    assert d.max() > 0 and d.max() < np.finfo(np.float32).tiny
    _check(ctx, d)                                  # stays denormal / lowest binades: exact integer arithmetic
    big = (rng.integers(0, 2 ** 20, size=(8, 30000)).astype(np.float32) * tiny).astype(np.float32)
    _check(ctx, big)                                # crosses from denormal into normal binades
    huge = np.full((4, 20000), np.float32(3e38), np.float32)
    huge[1, :] = np.float32(2e35)
    huge[2, :5000] = np.float32(0)
    huge[3] = rng.random(20000, dtype=np.float32) * np.float32(3e33)
    want = _ref(huge)
    assert np.isinf(want[0]) and np.isinf(want[1]) and np.isinf(want[2]) and np.isfinite(want[3])
    _same(ctx.sequential_sum(huge), want)


def test_chains_the_scan_does_not_take_fall_back_to_the_plain_loop(ctx):
    rng = np.random.default_rng(2)
    t = rng.standard_normal((8, 9000)).astype(np.float32)       # negative terms
    t[3] = np.abs(t[3])
    t[4, 8999] = -np.float32(0.0)                                # a negative zero at the very end
    t[5] = np.abs(t[5])
    t[5, 4500] = np.inf
    _check(ctx, t)
    n = np.abs(t[:2]).copy()
    n[0, 17] = np.nan
    got = ctx.sequential_sum(n)
    assert np.isnan(got[0]) and got[1].view(np.uint32) == _ref(n[1])[0].view(np.uint32)


def test_many_chains_of_em_size(ctx):
    """the shape of the EM's use: 40 cells of 4^9 weights, heavy-tailed like k-mer weights"""
    rng = np.random.default_rng(77)
    t = rng.lognormal(mean=-6.0, sigma=4.0, size=(40, 262144)).astype(np.float32)
    t[rng.random(t.shape) < 0.3] = 0
    _check(ctx, t)
