#!/usr/bin/env python3
"""Quick differential check of the two-launch serial EM (em_serial_scan = 3) against the three-launch scheme (2) and the
dependent-addition fold (0) on the GPU box: random tables, heavy tails, early stops, ragged batches, wrong estimates and
timed-out look-backs.  python tests/tools/em_fused_check.py [--W 10 12]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import peng_motif_amd as pk  # noqa: E402


def run(ctx, W, pwms, counts, bgd, thr, its, scan, skew=0, lookback=0, streams=2, budget=0, rows=1):
    ctx.set_option("em_fast", 2)
    ctx.test_em_generation(scan)
    ctx.set_option("em_test_skew", skew)
    ctx.set_option("em_test_lookback", lookback)
    ctx.set_option("em_overlap", streams)
    ctx.set_option("em_table_budget_mb", budget)
    try:
        t = time.time()
        out = ctx.em(W, pwms.copy(), counts, bgd, 1e4, thr, its)
        dt = time.time() - t
        met = {k: ctx.info("em_" + k) for k in ("fetched_blocks", "mispredicted_blocks")}
    finally:
        ctx.set_option("em_test_skew", 0)
        ctx.set_option("em_test_lookback", 0)
        ctx.set_option("em_overlap", 2)
        ctx.set_option("em_table_budget_mb", 0)
        ctx.test_em_generation(2)
        ctx.set_option("em_fast", 1)
    return out, dt, met


def same(a, b):
    return a[0].tobytes() == b[0].tobytes() and a[1].tolist() == b[1].tolist() and a[2].tobytes() == b[2].tobytes()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--W", type=int, nargs="+", default=[10, 12])
    a = ap.parse_args()
    ctx = pk.Context(0)
    bad = 0
    for W in a.W:
        NP = 4 ** W
        rng = np.random.default_rng(W)
        c = rng.lognormal(1.0, 2.5, NP).astype(np.uint32)
        bg = (rng.dirichlet(np.ones(64), size=NP // 64).reshape(-1)).astype(np.float32)
        counts, bgd = pk.DeviceArray.from_host(ctx, c), pk.DeviceArray.from_host(ctx, bg)
        n = 21 if W == 10 else 5
        pwms = np.maximum(rng.dirichlet(np.full(4, 0.5), size=(n, W)).astype(np.float32), np.float32(1e-20))
        ref0, _, _ = run(ctx, W, pwms, counts, bgd, 0.0, 3, 0)
        thr = float(np.median(ref0[2]))
        for its, th in ((3, 0.0), (6, thr), (1, 0.0), (0, 0.0), (2, 1e9)):
            ref, _, _ = run(ctx, W, pwms, counts, bgd, th, its, 0)
            for scan, skew, lb, streams, budget, rows in ((2, 0, 0, 2, 0, 1), (3, 0, 0, 2, 0, 1), (3, 0, 0, 2, 0, 0), (3, 0, 0, 1, 0, 1), (3, 2, 0, 2, 0, 1), (3, 0, 3, 2, 0, 1),
                                                   (3, 5, 7, 3, 0, 1), (3, 0, 1, 2, 0, 1), (3, 0, 0, 2, 32 if W == 10 else 256, 1), (3, 1, 0, 2, 0, 1),
                                                   (1, 0, 0, 2, 0, 1), (2, 3, 0, 1, 0, 1), (2, 0, 0, 3, 32 if W == 10 else 256, 1)):
                got, dt, met = run(ctx, W, pwms, counts, bgd, th, its, scan, skew, lb, streams, budget, rows)
                ok = same(got, ref)
                bad += not ok
                print("W=%d its=%d thr=%g scan=%d skew=%d lookback=%d streams=%d budget=%d rows=%d: %s  %.1f ms  %s iters=%s" % (
                    W, its, th, scan, skew, lb, streams, budget, rows, "ok" if ok else "DIFFERENT", dt * 1e3, met,
                    sorted(set(got[1].tolist()))), flush=True)
    # a degenerate PWM (zero entries -> 0/0 weights): the plain-loop path
    W = 10
    NP = 4 ** W
    rng = np.random.default_rng(3)
    c = rng.integers(0, 50, NP).astype(np.uint32)
    bg = np.full(NP, np.float32(1.0 / NP))
    bg[::977] = 0.0
    counts, bgd = pk.DeviceArray.from_host(ctx, c), pk.DeviceArray.from_host(ctx, bg)
    pwms = rng.dirichlet(np.ones(4), size=(3, W)).astype(np.float32)
    pwms[1, 3] = (0.0, 0.5, 0.5, 0.0)
    ref, _, _ = run(ctx, W, pwms, counts, bgd, 0.0, 2, 0)
    for scan in (3, 2):
        got, dt, met = run(ctx, W, pwms, counts, bgd, 0.0, 2, scan)
        ok = got[0].tobytes() == ref[0].tobytes() and got[1].tolist() == ref[1].tolist()
        bad += not ok
        print("degenerate W=10 scan=%d: %s %.1f ms nan=%d" % (scan, "ok" if ok else "DIFFERENT", dt * 1e3, int(np.isnan(got[0]).sum())), flush=True)
    ctx.close()
    print("FAILURES:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
