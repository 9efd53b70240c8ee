"""The arithmetic behind peng-motif_amd/csrc/seqsum.h, restated in numpy float32 and checked on the CPU.

seqsum.h evaluates a left-to-right float32 sum with a wave-wide scan (the serial mode of the EM).  Its claims -- inside one
binade the increment a stretch of terms gives depends on the parity of the running sum only; increments can be measured
from the binade's two lowest values; stretches compose; leaving the binade is never missed -- do not depend on the GPU.
This model follows the kernel step by step (rows, two bases, prefix composition, first flagged row added the plain way,
re-evaluation in the new binade) with small rows so that pure Python can run thousands of cases, and must reproduce
numpy's sequential float32 sum bit for bit.  (The device implementation itself is tested in tests/test_gpu_seqsum.py.)
"""
import numpy as np
import pytest

F = np.float32
ROWS, SEG = 8, 8  # the kernel: 64 rows of 64 terms


def bits(v):
    return int(np.float32(v).view(np.uint32))


def from_bits(b):
    return np.uint32(b).view(np.float32)


def bases_of(s):
    e = bits(s) >> 23
    if e <= 1:  # zero, denormals and the lowest normal binade: one "binade" with u = 2^-149
        return F(0), from_bits(1), 2 << 23
    return from_bits(e << 23), from_bits((e << 23) | 1), (e + 1) << 23


def compose(a, b, base):
    """(a then b): a_p + (b_q - base_q), q = parity of a_p"""
    d = (F(b[0] - base[0]), F(b[1] - base[1]))
    return tuple(F(a[p] + d[bits(a[p]) & 1]) for p in (0, 1))


def run(row, x):
    for t in row:
        x = F(x + t)
    return x


def fold_block(rows, s):
    first = 0
    while True:
        if bits(s) >= 0x7F800000:
            return s
        b0, b1, limit = bases_of(s)
        base = (b0, b1)
        inc = []  # per row: value reached from the even / odd base
        for l, row in enumerate(rows):
            inc.append(base if l < first else (run(row, b0), run(row, b1)))
        pre, acc = [], base
        for x in inc:  # inclusive prefix composition (the kernel does it in log steps; composition is associative)
            acc = compose(acc, x, base)
            pre.append(acc)
        par = bits(s) & 1
        ends = [F(s + F(p[par] - base[par])) for p in pre]
        flagged = [bits(e) >= limit for e in ends]
        if not any(flagged):
            return ends[-1]
        L = flagged.index(True)
        v = s if L == 0 else ends[L - 1]
        s = run(rows[L], v)  # the reference's own additions through the first row that leaves the binade
        first = L + 1
        if first == len(rows):
            return s


def model_sum(t):
    t = np.asarray(t, np.float32)
    pad = (-len(t)) % (ROWS * SEG)
    t = np.concatenate([t, np.zeros(pad, np.float32)])
    s = F(0)
    with np.errstate(over="ignore", invalid="ignore"):
        for b in range(0, len(t), ROWS * SEG):
            s = fold_block(t[b:b + ROWS * SEG].reshape(ROWS, SEG), s)
    return s


def ref_sum(t):
    t = np.asarray(t, np.float32)
    if len(t) == 0:
        return F(0)
    with np.errstate(over="ignore", invalid="ignore"):
        return np.cumsum(t, dtype=np.float32)[-1]


def check(t):
    with np.errstate(all="ignore"):
        assert bits(model_sum(t)) == bits(ref_sum(t))


def test_composition_is_associative_on_in_binade_stretches():
    rng = np.random.default_rng(1)
    b0, b1, limit = bases_of(F(1.5))
    base = (b0, b1)
    for _ in range(300):
        xs = []
        for _ in range(3):
            row = (rng.integers(0, 8, 6) * 2.0 ** -24).astype(np.float32)  # half-ulps of the binade of 1.0: ties galore
            with np.errstate(all="ignore"):
                xs.append((run(row, b0), run(row, b1)))
        left = compose(compose(xs[0], xs[1], base), xs[2], base)
        right = compose(xs[0], compose(xs[1], xs[2], base), base)
        assert [bits(v) for v in left] == [bits(v) for v in right]


@pytest.mark.parametrize("seed", range(8))
def test_model_equals_sequential_sum_on_random_chains(seed):
    rng = np.random.default_rng(seed)
    for _ in range(60):
        n = int(rng.integers(1, 600))
        kind = rng.integers(0, 6)
        if kind == 0:
            t = rng.random(n, dtype=np.float32)
        elif kind == 1:    # the whole exponent range
            with np.errstate(over="ignore", under="ignore"):
                t = np.ldexp(1.0 + rng.random(n), rng.integers(-149, 120, n)).astype(np.float32)
            t[~np.isfinite(t)] = 0
        elif kind == 2:    # ties: a head, then multiples of its half-ulp
            head = F(rng.choice([1.0, 1.0 + 2.0 ** -23, 3.0, 2.0 ** 20]))
            t = (rng.integers(0, 5, n) * np.spacing(head) / 2).astype(np.float32)
            t[0] = head
        elif kind == 3:    # sparse, with giants
            t = np.zeros(n, np.float32)
            idx = rng.integers(0, n, max(1, n // 10))
            t[idx] = (rng.random(idx.size) * 10.0 ** rng.integers(-8, 8, idx.size)).astype(np.float32)
        elif kind == 4:    # denormals growing into the normal range
            t = (rng.integers(0, 2 ** 18, n).astype(np.float32) * F(1e-45)).astype(np.float32)
        else:              # doubling sums: a crossing in every row
            with np.errstate(over="ignore"):
                t = np.ldexp(1.0, np.arange(n) // int(rng.integers(1, 9))).astype(np.float32)
            t[~np.isfinite(t)] = 0
        check(t)


def test_overflow_and_zero_chains():
    check(np.zeros(500, np.float32))
    check(np.full(300, 3e38, np.float32))          # +inf absorbs the rest
    check(np.concatenate([np.zeros(70, np.float32), np.full(300, 2e36, np.float32)]))
    check([])


# ---- the blocks evaluated AHEAD of their chain (seqsum.h: block_increments, walk_chain; em.hip: block_binade) ------------
NO_BINADE = 0xFFFFFFFF


def binade_of(s):
    e = bits(s) >> 23
    return 1 if e <= 1 else e


def bases_of_binade(e):
    return from_bits(0 if e <= 1 else e << 23), from_bits(1 if e <= 1 else (e << 23) | 1), (e + 1) << 23


def block_binade(before, after, wrong=False):
    """the binade both estimates share, widened by 2^-9 -- or NO_BINADE; `wrong`: the test hook's sabotage"""
    lo, hi = F(before * (1.0 - 1.0 / 512.0)), F(after * (1.0 + 1.0 / 512.0))
    sane = lo >= 0 and hi < from_bits(0x7F000000)
    e = binade_of(lo) if sane and binade_of(lo) == binade_of(hi) else NO_BINADE
    if wrong and sane:
        e = binade_of(lo) if e == NO_BINADE else (e + 1 if e < 200 else e)
    return e


def block_increments(rows, e):
    """both bases of binade e run through the block: the two increments, or None if a base chain leaves the binade"""
    b0, b1, limit = bases_of_binade(e)
    base, acc = (b0, b1), (b0, b1)
    for row in rows:
        acc = compose(acc, (run(row, b0), run(row, b1)), base)
    if bits(acc[0]) >= limit or bits(acc[1]) >= limit:
        return None
    return F(acc[0] - b0), F(acc[1] - b1)


def ahead_sum(t, wrong_every=0):
    """what em_weights_span / em_span_eval / em_chain do to one cell: plain block sums -> estimates -> a binade per block ->
    increments of the blocks that got one (all at once, no chain needed) -> the chain: runs of blocks that share the
    binade of s compose (one addition per run here: the model composes them one by one), everything else goes to
    fold_block.  Returns (sum, blocks that took the short way)."""
    t = np.asarray(t, np.float32)
    pad = (-len(t)) % (ROWS * SEG)
    t = np.concatenate([t, np.zeros(pad, np.float32)])
    blocks = [t[b:b + ROWS * SEG].reshape(ROWS, SEG) for b in range(0, len(t), ROWS * SEG)]
    with np.errstate(all="ignore"):
        sums = [np.float32(np.sum(b.astype(np.float64))) for b in blocks]  # (any order: an estimate)
        rec, before = [], 0.0
        for i, (b, sb) in enumerate(zip(blocks, sums)):
            e = block_binade(before, before + float(sb), wrong=bool(wrong_every) and i % wrong_every == 0)
            inc = block_increments(b, e) if e != NO_BINADE else None
            rec.append((e, inc) if inc is not None else (NO_BINADE, None))
            before += float(sb)
        s, short = F(0), 0
        for b, (e, inc) in zip(blocks, rec):
            if e != NO_BINADE and bits(s) < 0x7F800000 and binade_of(s) == e:
                s2 = F(s + inc[bits(s) & 1])
                if bits(s2) < (e + 1) << 23:   # the test fold_block applies to its last row
                    s, short = s2, short + 1
                    continue
            s = fold_block(b, s)
        return s, short


@pytest.mark.parametrize("seed", range(6))
def test_blocks_ahead_equal_the_sequential_sum(seed):
    rng = np.random.default_rng(100 + seed)
    took_short = 0
    for _ in range(40):
        n = int(rng.integers(200, 3000))
        kind = rng.integers(0, 5)
        if kind == 0:      # EM-like: heavy-tailed non-negative weights
            t = rng.lognormal(0.0, 2.5, n).astype(np.float32)
        elif kind == 1:    # ties: multiples of half an ulp of the running sum's binade
            t = (rng.integers(0, 5, n) * 2.0 ** -24).astype(np.float32)
            t[0] = 1.0
        elif kind == 2:    # sparse with giants
            t = np.zeros(n, np.float32)
            idx = rng.integers(0, n, max(1, n // 10))
            t[idx] = (rng.random(idx.size) * 10.0 ** rng.integers(-8, 8, idx.size)).astype(np.float32)
        elif kind == 3:    # long flat stretch: the serial sum stagnates (terms below half an ulp) while the estimate grows
            t = np.full(n, 2.0 ** -26, np.float32)
            t[0] = 1.0
        else:
            t = rng.random(n, dtype=np.float32)
        for wrong_every in (0, 1, 3):
            with np.errstate(all="ignore"):
                got, short = ahead_sum(t, wrong_every)
            assert bits(got) == bits(ref_sum(t)), (seed, kind, wrong_every)
            took_short += short if wrong_every == 0 else 0
    assert took_short > 100   # the short way is the common one


def test_blocks_ahead_never_trust_the_estimate():
    """A chain whose serial sum stagnates just below 2 (every term is half an ulp: a tie, rounded to the even neighbour,
    which is the sum itself) while its exact sum -- the estimate -- walks on past 2 and out of the margin: the estimate
    hands the later blocks the binade of 2, the chain's sum is still in the binade of 1, the check turns the increments
    down and fold_block reproduces the stagnation."""
    n = 64 * 4000
    t = np.full(n, 2.0 ** -24, np.float32)
    t[0] = 2.0 - 2.0 ** -22   # even mantissa
    exact = float(np.sum(t.astype(np.float64)))
    assert exact > 2.0 * (1 + 1 / 256) and bits(ref_sum(t)) == bits(F(t[0]))
    with np.errstate(all="ignore"):
        got, short = ahead_sum(t)
    assert bits(got) == bits(F(t[0]))
    # and the estimate did mislead: blocks were evaluated for the binade of 2 that the chain never entered
    pre = np.concatenate([[0.0], np.cumsum(t.astype(np.float64).reshape(-1, ROWS * SEG).sum(axis=1))])
    b2 = binade_of(F(2.0))
    assert sum(block_binade(pre[i], pre[i + 1]) == b2 for i in range(len(pre) - 1)) > 500
