"""Differential soak of the serial EM: the library's scheme (blocks evaluated ahead of their chain, 2), its two-launch variant
(3) and the scan block after block (generation 1) against the dependent-addition fold (generation 0) on random count
tables, backgrounds and PWMs (GPU box) -- PWMs, iteration counts and `change` must be equal bit for bit.  Wrong binade
estimates (em_test_skew) and timed-out look-backs (em_test_lookback) are part of the mix.
usage: python tests/tools/em_scan_fuzz.py FIRST_SEED SECONDS [W,W,...]   (default 8,10,10)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import peng_motif_amd as pk

seed0, seconds = int(sys.argv[1]), float(sys.argv[2])
WS = [int(w) for w in sys.argv[3].split(",")] if len(sys.argv) > 3 else [8, 10, 10]
ctx = pk.Context(0)
ctx.set_option("em_fast", 2)
t_end = time.time() + seconds
seed, n_pwm_total, mismatches = seed0, 0, 0
t_print = time.time()
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    W = int(rng.choice(WS))
    NP = 4 ** W
    kind = int(rng.integers(0, 4))
    if kind == 0:      # sparse counts (a small input)
        c = np.zeros(NP, np.uint32)
        idx = rng.integers(0, NP, NP // int(rng.integers(4, 200)))
        c[idx] = rng.integers(1, 50, idx.size)
    elif kind == 1:    # dense, heavy-tailed
        c = rng.lognormal(2.0, 2.0, NP).astype(np.uint32)
    elif kind == 2:    # huge counts on a few k-mers
        c = rng.integers(0, 4, NP).astype(np.uint32)
        c[rng.integers(0, NP, 20)] = rng.integers(10 ** 5, 10 ** 8, 20)
    else:
        c = rng.integers(0, 2000, NP).astype(np.uint32)
    bg = (rng.dirichlet(np.ones(64), size=NP // 64).reshape(-1) * rng.choice([1.0, 1e-3, 1e-6])).astype(np.float32)
    n = int(rng.integers(1, 20))
    conc = float(rng.choice([0.05, 0.3, 1.0, 5.0]))   # small concentration: PWM entries down to 1e-30
    pw = rng.dirichlet(np.full(4, conc), size=(n, W)).astype(np.float32)
    pw = np.maximum(pw, np.float32(1e-30))
    sat, thr, it = float(rng.choice([1e4, 1e3, 1e5])), float(rng.choice([0.0, 0.08])), int(rng.integers(1, 6))
    counts = pk.DeviceArray.from_host(ctx, c)
    bgd = pk.DeviceArray.from_host(ctx, bg)
    out = {}
    skew = int(rng.choice([0, 0, 1, 3, 17]))   # test hook: wrong binade estimates for every skew-th block (mode 2 only)
    lookback = int(rng.choice([0, 0, 1, 5]))
    ctx.set_option("em_head_blocks", int(rng.choice([1, 1, 2, 4, 9])))  # (blocks folded from zero beside the evaluation: scheme 2)
    for scan in (3, 2, 1, 0):
        ctx.test_em_generation(scan)
        ctx.set_option("em_test_skew", skew if scan >= 2 else 0)
        ctx.set_option("em_test_lookback", lookback if scan == 3 else 0)
        out[scan] = ctx.em(W, pw, counts, bgd, sat, thr, it)
    ctx.test_em_generation(2)
    ctx.set_option("em_test_skew", 0)
    ctx.set_option("em_test_lookback", 0)
    ctx.set_option("em_head_blocks", 1)
    bad_modes = [k for k in (1, 2, 3) if not (out[k][0].tobytes() == out[0][0].tobytes() and out[k][1].tolist() == out[0][1].tolist()
                                           and out[k][2].tobytes() == out[0][2].tobytes())]
    if bad_modes:
        # once more, all three: which of them moves?
        again = {}
        for scan in (3, 2, 1, 0):
            ctx.test_em_generation(scan)
            again[scan] = ctx.em(W, pw, counts, bgd, sat, thr, it)
        ctx.test_em_generation(2)
        print("MISMATCH seed", seed, "W", W, "kind", kind, "skew", skew, "lookback", lookback, "n", n, "it", it, "thr", thr, "modes that differ from the fold:", bad_modes,
              "; repeated run equals first run per mode:", {k: again[k][0].tobytes() == out[k][0].tobytes() for k in (3, 2, 1, 0)})
        mismatches += 1
        if mismatches >= 5:
            sys.exit(1)
    n_pwm_total += n
    seed += 1
    if time.time() - t_print > 60:   # (a sign of life: the GPU pool takes a silent command for a hung one)
        print("... seed", seed, "PWMs", n_pwm_total, "mismatches", mismatches, flush=True)
        t_print = time.time()
if mismatches:
    sys.exit(1)
print("EM scan fuzz: seeds %d..%d, %d PWMs, blocks ahead (three launches / two launches) == block after block == dependent fold bit for bit" % (seed0, seed - 1, n_pwm_total))
