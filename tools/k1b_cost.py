import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, peng_motif_amd as pk
ctx = pk.Context(0)
ctx.synth(1, 0, 10_000_000, 200, 10)
c = ctx.empty(4 ** 10, np.uint32); lt = ctx.empty(1, np.uint64); bg = ctx.empty(84, np.uint64)
t0, t1 = ctx.timer(), ctx.timer()
for name, fn in (("count_bg", lambda: ctx.count_bg(True, c, lt, bg)), ("count", lambda: ctx.count(True, c, lt)), ("count_bg", lambda: ctx.count_bg(True, c, lt, bg)), ("count", lambda: ctx.count(True, c, lt))):
    for _ in range(3): fn()
    ms = []
    for _ in range(15):
        ctx.record(t0); fn(); ctx.record(t1); ms.append(ctx.elapsed_ms(t0, t1))
    ms.sort(); print(name, "median %.4f ms  min %.4f" % (ms[len(ms)//2], ms[0]))
