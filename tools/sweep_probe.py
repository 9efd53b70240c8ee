#!/usr/bin/env python3
"""The K2+K3 sweep on its own, for profiling and A/B (GPU box):  python tools/sweep_probe.py [--W 12] [--impl 0|1] [--reps 20]
Counts a synthetic set, then times pengk_pattern_stats (HIP events), best and median of `reps`; --pairs = option sweep_pairs
(1: a pattern and its reverse complement evaluated once from W = 12 on (stats_pair_kernel), 0: one thread per pattern)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import peng_motif_amd as pk  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--W", type=int, default=12)
    ap.add_argument("--pairs", type=int, default=1)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--nseq", type=int, default=2_000_000)
    a = ap.parse_args()
    W = a.W
    ctx = pk.Context(0)
    ctx.synth(1, 0, a.nseq, 200, W)
    counts, ltot, bg = ctx.count_bg(True)
    ctx.mirror(W, counts)
    V = ctx.bg_model(bg, 2)
    ctx.set_option("sweep_pairs", a.pairs)
    out = ctx.pattern_stats(W, True, 2, 2, V, ltot, counts)
    t0, t1 = ctx.timer(), ctx.timer()
    ms = []
    for _ in range(a.reps):
        ctx.record(t0)
        ctx.pattern_stats(W, True, 2, 2, V, ltot, counts, *out)
        ctx.record(t1)
        ms.append(ctx.elapsed_ms(t0, t1))
    ms.sort()
    b = 28 * 4 ** W
    print("W=%d pairs=%d nseq=%d: best %.4f ms  median %.4f ms  = %.0f GB/s of 28 B per pattern" % (W, a.pairs, a.nseq, ms[0], ms[len(ms) // 2], b / ms[len(ms) // 2] / 1e6), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
