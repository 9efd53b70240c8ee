#!/usr/bin/env python3
"""Where the microseconds of the blocks-ahead EM's kernels go, workgroup by workgroup (GPU box, developer build):
    PENGK_BUILD_OUT=$PWD/ablation_libs/wgtrace.so PENGK_EXTRA_FLAGS=-DPENGK_WG_TRACE python peng-motif_amd/build.py --force
    PENGK_LIB=$PWD/ablation_libs/wgtrace.so python tools/em_wgtrace.py [--W 10] [--pwms 16] [--streams 1] [--iter 3]
With -DPENGK_WG_TRACE every workgroup of em_weights_span_kernel / em_span_eval_kernel (or em_span_fused_kernel) / em_chain_store_kernel records its start
and end (s_memtime) and its XCC (csrc/em.hip, WgTrace).  One pengk_em call is traced; the launches of iteration `--iter`
are printed: when the workgroups started (how many rounds a kernel really takes), how long they ran, and for the
evaluation how long a span was on its way."""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import peng_motif_amd as pk  # noqa: E402

NAMES = {1: "weights", 2: "evaluation", 3: "chains"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--W", type=int, default=10)
    ap.add_argument("--pwms", type=int, default=16)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--iter", type=int, default=3)
    ap.add_argument("--streams", type=int, default=1)
    ap.add_argument("--nseq", type=int, default=2_000_000)
    ap.add_argument("--scan", type=int, default=3)
    a = ap.parse_args()
    W, NP = a.W, 4 ** a.W
    ctx = pk.Context(0)
    lib = pk.lib()
    raw = ctypes.CDLL(pk.LIB_PATH)
    raw.pengk_debug_wg_trace.restype = ctypes.c_longlong
    raw.pengk_debug_wg_trace.argtypes = [ctypes.c_void_p, ctypes.c_uint]
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
    out = ctx.empty(pw.shape, np.float32)
    state = pk.DeviceArray.from_host(ctx, np.zeros((a.pwms, 2), np.int32))
    change = pk.DeviceArray.from_host(ctx, np.zeros(a.pwms, np.float32))
    ctx.set_option("em_fast", 2)
    ctx.set_option("em_overlap", a.streams)
    ctx.test_em_generation(a.scan)
    t0, t1 = ctx.timer(), ctx.timer()
    MAX = 1 << 20
    buf = np.zeros(3 * MAX, np.uint64)
    for r in range(3):
        pk._check(lib.pengk_memcpy_h2d(ctx.h, out.ptr, pw.ctypes.data, pw.nbytes))
        ctx.synchronize()
        raw.pengk_debug_wg_trace(buf.ctypes.data, 0)  # start over
        ctx.record(t0)
        ctx.em_device(W, a.pwms, out, counts, bgk, state, change, 1e4, 0.0, a.iters)
        ctx.record(t1)
        ms = ctx.elapsed_ms(t0, t1)
    n = raw.pengk_debug_wg_trace(buf.ctypes.data, MAX)
    rec = buf[: 3 * min(n, MAX)].reshape(-1, 3)
    head, s, e = rec[:, 0], rec[:, 1].astype(np.int64), rec[:, 2].astype(np.int64)
    kern, kind, xcc, wg = (head >> 56).astype(int), ((head >> 48) & 255).astype(int), ((head >> 40) & 255).astype(int), (head & ((1 << 40) - 1)).astype(int)
    if os.environ.get("PENGK_WGTRACE_DEBUG"):
        for x in sorted(set(xcc.tolist())):
            mm = xcc == x
            print("xcc", x, "records", int(mm.sum()), "min start", int(s[mm].min()), "max end", int(e[mm].max()), "span", int(e[mm].max() - s[mm].min()))
    span_ticks = int(e.max() - s.min())
    tick_us = ms * 1e3 / span_ticks  # (the call's first workgroup to its last against the events around it: a little generous)
    print("W=%d, %d PWMs x %d iterations on %d stream(s): %.1f us by the events, %d records, one tick ~ %.5f us" % (W, a.pwms, a.iters, a.streams, ms * 1e3, n, tick_us))
    T0 = int(s.min())
    for k in (1, 2, 3):
        m = (kern == k) & ~((k == 2) & ((kind == 2) | (kind == 3) | (kind == 4)))
        if not m.any():
            continue
        # the launches of this kernel: its workgroups' starts, sorted, split where nothing starts for a while
        order = np.argsort(s[m])
        ss, ee, kk, ww = s[m][order], e[m][order], kind[m][order], wg[m][order]
        cuts = [0] + [i for i in range(1, len(ss)) if ss[i] - ee[:i].max() > 0] + [len(ss)]
        launches = [(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1)]
        print("%s: %d launches seen" % (NAMES[k], len(launches)))
        if a.iter >= len(launches):
            continue
        lo, hi = launches[a.iter]
        s0 = ss[lo]
        dur = (ee[lo:hi] - ss[lo:hi]) * tick_us
        st = (ss[lo:hi] - s0) * tick_us
        print("  launch %d: %d workgroups, first start to last end %.1f us (starts %.1f us after the call's first workgroup)" % (
            a.iter, hi - lo, (ee[lo:hi].max() - s0) * tick_us, (s0 - T0) * tick_us))
        print("    a workgroup runs %.1f us on average (median %.1f, longest %.1f, shortest %.1f)" % (dur.mean(), np.median(dur), dur.max(), dur.min()))
        hist = np.histogram(st, bins=[0, 1, 2, 4, 6, 8, 10, 12, 15, 20, 25, 30, 40, 1e9])[0]
        print("    starts by time since the first [0,1,2,4,6,8,10,12,15,20,25,30,40,..) us:", hist.tolist())
        ends = (ee[lo:hi] - s0) * tick_us
        print("    ends: 50 %% of the workgroups by %.1f us, 90 %% by %.1f, 99 %% by %.1f, all by %.1f" % tuple(np.percentile(ends, [50, 90, 99, 100])))
        if k == 2:
            ex = kk[lo:hi] == 1
            if ex.any():
                print("    block-0 workgroups: %d, run %.1f us on average (longest %.1f), the last ends at %.1f us" % (ex.sum(), dur[ex].mean(), dur[ex].max(), ends[ex].max()))
            for kd, what in ((4, "the head (previous iteration's finalize step) is done"), (3, "thread 0's weights are done"), (2, "a span is in LDS (and its estimates known)")):
                m2 = (kern == 2) & (kind == kd) & (s >= s0) & (s <= ee[lo:hi].max())
                if m2.any():
                    ld = (e[m2] - s[m2]) * tick_us
                    print("    %s %.1f us after its workgroup started (median %.1f, longest %.1f)" % (what, ld.mean(), np.median(ld), ld.max()))
        if k == 3:
            f = kk[lo:hi]
            nf = f[f < 254]
            dd = dur[f < 254]
            if len(nf):
                i = int(np.argmax(dd))
                print("    the longest chain took %d blocks the long way (average %.1f); chains with <= 4 such blocks run %.1f us, with >= 12: %.1f us" % (
                    nf[i], nf.mean(), dd[nf <= 4].mean() if (nf <= 4).any() else float('nan'), dd[nf >= 12].mean() if (nf >= 12).any() else float('nan')))
    ctx.close()


if __name__ == "__main__":
    main()
