"""Soak of the three-level partitioned count of W = 14 against the direct (one atomic per window) emitter, which
tests/test_gpu_parity.py pins to the oracle at W = 14: random inputs as in the K1 fuzz (ragged and too-short sequences,
invalid bases, low-complexity stretches, split items), both strand modes, the fused background counters, and the slices
of every level forced to overflow on a third of the cases.  GPU box.  usage: python tests/tools/count_w14_fuzz.py FIRST_SEED SECONDS"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import peng_motif_amd as pk

W = 14
seed0, seconds = int(sys.argv[1]), float(sys.argv[2])
ctx = pk.Context(0)
t_end = time.time() + seconds
seed = seed0
while time.time() < t_end:
    rng = np.random.default_rng(140000 + seed)
    both = bool(rng.integers(0, 2))
    M = int(rng.choice([0, 64, 100, 256]))
    cap = int(rng.choice([0, 0, 64, 128]))
    n_seq = int(rng.integers(1, 3000))
    p_invalid = float(rng.choice([0.0, 0.0, 1e-3, 2e-2]))
    seqs = []
    for _ in range(n_seq):
        kind = rng.integers(0, 10)
        L = int(rng.integers(1, W)) if kind == 0 else int(rng.integers(W, 60)) if kind < 3 else int(rng.integers(60, 700)) \
            if kind < 9 else int(rng.integers(2000, 6000))
        s = rng.integers(1, 5, size=L).astype(np.uint8)
        if rng.random() < 0.3 and L > 40:
            unit = rng.integers(1, 5, size=int(rng.integers(1, 16))).astype(np.uint8)
            a = int(rng.integers(0, L - 20))
            b = min(L, a + int(rng.integers(20, 400)))
            s[a:b] = np.tile(unit, (b - a) // len(unit) + 1)[:b - a]
        if p_invalid:
            s[rng.random(L) < p_invalid] = 0
        seqs.append(s)
    if rng.random() < 0.2:  # many lanes of a wave on one bucket
        seqs = seqs[:5] + [seqs[0]] * 200
    codes = np.concatenate(seqs)
    offs = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.int64)
    p = pk.Packed(codes, offs, W, M)
    ctx.upload(p)
    res = []
    for impl in (1, 2):
        ctx.set_option("count_impl", impl)
        ctx.set_option("key_cap_override", cap if impl == 2 else 0)
        if p.all_whole:
            counts, lt, bg = ctx.count_bg(both)
            res.append((counts.to_host().copy(), int(lt.to_host()[0]), bg.to_host().copy()))
        else:
            counts, lt = ctx.count(both)
            res.append((counts.to_host().copy(), int(lt.to_host()[0]), None))
    ctx.set_option("count_impl", 0)
    ctx.set_option("key_cap_override", 0)
    ok = res[0][1] == res[1][1] and np.array_equal(res[0][0], res[1][0]) and (res[0][2] is None or np.array_equal(res[0][2], res[1][2]))
    if res[0][2] is not None:
        ok = ok and np.array_equal(res[1][2].astype(np.int64), p.bg_counts)
    if not ok:
        print("MISMATCH seed", seed, "both", both, "M", M, "cap", cap, "bins differing", int((res[0][0] != res[1][0]).sum()))
        sys.exit(1)
    seed += 1
print("W=14 count fuzz: seeds %d..%d (key_cap_override 0 / 64 / 128), three-level partition == direct emitter bit for bit" % (seed0, seed - 1))
