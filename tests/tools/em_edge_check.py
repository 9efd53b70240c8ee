"""Developer check (GPU): the serial EM's iteration protocol at its edges -- 0 / 1 / 2 / 11 iterations, thresholds that end
every PWM at once, some PWMs early, none; 1 PWM, 9 PWMs (a partial group of 8), many PWMs over several batches -- the
library's scheme against the dependent-addition fold (pengk_test_em_generation 0), bit for bit, and repeated calls on one
context (the parity double-buffering must not carry flags from call to call)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import peng_motif_amd as pk

ctx = pk.Context(0)
fails = 0
for W in (10, 12):
    NP = 4 ** W
    rng = np.random.default_rng(W)
    c = rng.poisson(2.0, NP).astype(np.uint32)
    bg = (rng.dirichlet(np.ones(64), size=NP // 64).reshape(-1)).astype(np.float32)
    counts, bgd = pk.DeviceArray.from_host(ctx, c), pk.DeviceArray.from_host(ctx, bg)
    ctx.set_option("em_fast", 2)
    for n in ((1, 9, 40) if W == 10 else (1, 9)):
        pw = rng.dirichlet(np.ones(4) * 2, size=(n, W)).astype(np.float32)
        pw[0] = np.float32(0.25)  # (uniform rows: converges at once)
        for max_it in (0, 1, 2, 11):
            for thr in (0.0, 0.05, 1e9):
                if W == 12 and max_it == 11:
                    continue
                for budget in ((0, 20) if n == 40 else (0,)):
                    ctx.set_option("em_table_budget_mb", budget)
                    ctx.test_em_generation(0)
                    ref = ctx.em(W, pw, counts, bgd, 1e4, thr, max_it)
                    ctx.test_em_generation(2)
                    a = ctx.em(W, pw, counts, bgd, 1e4, thr, max_it)
                    b = ctx.em(W, pw, counts, bgd, 1e4, thr, max_it)  # again: no state may survive a call
                    ok = all(x[0].tobytes() == ref[0].tobytes() and x[1].tolist() == ref[1].tolist() and x[2].tobytes() == ref[2].tobytes() for x in (a, b))
                    if not ok:
                        fails += 1
                        print("MISMATCH W=%d n=%d max_it=%d thr=%g budget=%d: iters %s / %s" % (W, n, max_it, thr, budget, a[1].tolist()[:6], ref[1].tolist()[:6]))
        print("W=%d n=%d done; iterations seen at thr 0.05 / 11: %s" % (W, n, sorted(set(ctx.em(W, pw, counts, bgd, 1e4, 0.05, 11)[1].tolist()))), flush=True)
    ctx.set_option("em_table_budget_mb", 0)
print("FAILURES:", fails)
sys.exit(1 if fails else 0)
