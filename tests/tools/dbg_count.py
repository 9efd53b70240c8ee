import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import peng_motif_amd as pk
from oracle import oracle as po
ctx = pk.Context(0)
for (W, n, L, both) in [(10, 64, 200, False), (10, 300, 200, False), (10, 5000, 205, False), (10, 5000, 205, True), (8, 5000, 205, True)]:
    codes, offs = po.synth(3, 0, n, L)
    want, lt = po.count(codes, offs, W, both)
    if both:
        # un-mirror: keep canonical only
        pass
    p = pk.Packed(codes, offs, W)
    ctx.upload(p)
    for impl in (1, 2):
        ctx.set_option("count_impl", impl)
        c, l = ctx.count(both)
        if both: ctx.mirror(W, c)
        got = c.to_host().astype(np.uint64)
        bad = np.nonzero(got != want)[0]
        print("W", W, "n", n, "both", both, "impl", impl, "sum got/want", int(got.sum()), int(want.sum()), "nbad", len(bad))
        if len(bad):
            nb = 2*W-15
            print("  bad buckets", np.unique(bad & ((1<<nb)-1))[:40], " first bad", bad[:10], got[bad[:10]], want[bad[:10]])
            d = got.astype(np.int64) - want.astype(np.int64)
            print("  diff per bucket", [int(d[(np.arange(4**W) & ((1<<nb)-1)) == b].sum()) for b in range(min(1<<nb, 32))])
