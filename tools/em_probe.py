#!/usr/bin/env python3
"""The parity EM on its own, for profiling and A/B (GPU box):  python tools/em_probe.py [--W 10] [--pwms 16] [--reps 20]
  [--streams N] [--nseq 2000000]
Counts a synthetic set, then times pengk_em_device in the serial mode (em_fast = 2) on `pwms` seed PWMs x 10 iterations,
best and median of `reps`, and prints what the chains met (pengk_get_info em_*).  Under `rocprofv3 --kernel-trace` the
kernel timeline of these calls shows the gaps between the dependent launches (tools/em_timeline.py reads the trace)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import peng_motif_amd as pk  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--W", type=int, default=10)
    ap.add_argument("--pwms", type=int, default=16)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--streams", type=int, default=0)
    ap.add_argument("--nseq", type=int, default=2_000_000)
    ap.add_argument("--fast", type=int, default=2)
    ap.add_argument("--lean", type=int, default=1, help="option em_lean_div")
    ap.add_argument("--head", type=int, default=-1, help="option em_head_blocks")
    ap.add_argument("--budget", type=int, default=-1, help="option em_table_budget_mb")
    ap.add_argument("--scan", type=int, default=2, help="pengk_test_em_generation (2 = the library's scheme, 3 = two launches per iteration, 1 = scan, 0 = fold)")
    a = ap.parse_args()
    W, NP = a.W, 4 ** a.W
    ctx = pk.Context(0)
    lib = pk.lib()
    ctx.synth(1, 0, a.nseq, 200, W)
    counts, ltot, bg = ctx.count_bg(True)
    ctx.mirror(W, counts)
    V = ctx.bg_model(bg, 2)
    bgprob, expected, logp, z = ctx.pattern_stats(W, True, 2, 2, V, ltot, counts)
    bgk = pk.DeviceArray.from_host(ctx, bgprob.to_host()[2])
    rng = np.random.default_rng(5)
    ids = rng.integers(0, NP, size=a.pwms)
    pw = np.full((a.pwms, W, 4), 0.1, np.float32)
    for j, x in enumerate(ids):
        for q in range(W):
            pw[j, q, (int(x) >> (2 * q)) & 3] = 0.7
    init = pk.DeviceArray.from_host(ctx, pw)
    out = ctx.empty(pw.shape, np.float32)
    state = pk.DeviceArray.from_host(ctx, np.zeros((a.pwms, 2), np.int32))
    change = pk.DeviceArray.from_host(ctx, np.zeros(a.pwms, np.float32))
    ctx.set_option("em_fast", a.fast)
    ctx.set_option("em_lean_div", a.lean)
    ctx.test_em_generation(a.scan)
    if a.head >= 0:
        ctx.set_option("em_head_blocks", a.head)
    if a.budget >= 0:
        ctx.set_option("em_table_budget_mb", a.budget)
    if a.streams:
        ctx.set_option("em_overlap", a.streams)
    t0, t1 = ctx.timer(), ctx.timer()
    ms = []
    for r in range(a.reps + 2):
        pk._check(lib.pengk_memcpy_h2d(ctx.h, out.ptr, pw.ctypes.data, pw.nbytes))
        ctx.record(t0)
        ctx.em_device(W, a.pwms, out, counts, bgk, state, change, 1e4, 0.0, a.iters)
        ctx.record(t1)
        ms.append(ctx.elapsed_ms(t0, t1))
    ms = sorted(ms[2:])
    met = {k: ctx.info("em_" + k) for k in ("fetched_blocks", "mispredicted_blocks", "restaged_blocks", "restaged_waits")} if a.fast == 2 else {}
    print("scan=%d streams=%d head=%d budget=%d " % (a.scan, a.streams, a.head, a.budget), end="")
    print("W=%d pwms=%d iters=%d: best %.4f ms  median %.4f ms  (%.2f us per iteration)  %s" % (
        W, a.pwms, a.iters, ms[0], ms[len(ms) // 2], ms[len(ms) // 2] * 1e3 / a.iters, met), flush=True)
    import hashlib
    try:
        print("sha256 of the PWMs", hashlib.sha256(out.to_host().tobytes()).hexdigest()[:16])
    except BrokenPipeError:  # (`| head -1`)
        pass
    ctx.close()


if __name__ == "__main__":
    main()
