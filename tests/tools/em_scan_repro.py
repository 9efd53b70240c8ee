import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import peng_motif_amd as pk
seed = int(sys.argv[1]); WS=[int(w) for w in sys.argv[2].split(",")]
ctx = pk.Context(0)
ctx.set_option("em_fast", 2)
rng = np.random.default_rng(seed)
W = int(rng.choice(WS)); NP = 4 ** W
kind = int(rng.integers(0, 4))
if kind == 0:
    c = np.zeros(NP, np.uint32); idx = rng.integers(0, NP, NP // int(rng.integers(4, 200))); c[idx] = rng.integers(1, 50, idx.size)
elif kind == 1:
    c = rng.lognormal(2.0, 2.0, NP).astype(np.uint32)
elif kind == 2:
    c = rng.integers(0, 4, NP).astype(np.uint32); c[rng.integers(0, NP, 20)] = rng.integers(10 ** 5, 10 ** 8, 20)
else:
    c = rng.integers(0, 2000, NP).astype(np.uint32)
bg = (rng.dirichlet(np.ones(64), size=NP // 64).reshape(-1) * rng.choice([1.0, 1e-3, 1e-6])).astype(np.float32)
n = int(rng.integers(1, 20)); conc = float(rng.choice([0.05, 0.3, 1.0, 5.0]))
pw = rng.dirichlet(np.full(4, conc), size=(n, W)).astype(np.float32); pw = np.maximum(pw, np.float32(1e-30))
sat, thr, it = float(rng.choice([1e4, 1e3, 1e5])), float(rng.choice([0.0, 0.08])), int(rng.integers(1, 6))
print("W", W, "kind", kind, "n", n, "conc", conc, "sat", sat, "thr", thr, "it", it)
counts = pk.DeviceArray.from_host(ctx, c); bgd = pk.DeviceArray.from_host(ctx, bg)
out = {}
for tag, scan, ov in (("fold",0,2),("scan1",1,2),("ahead_1stream",2,1),("ahead_2streams",2,2),("ahead_again",2,2)):
    ctx.test_em_generation(scan); ctx.set_option("em_overlap", ov)
    out[tag] = ctx.em(W, pw, counts, bgd, sat, thr, it)
for tag in out:
    if tag == "fold": continue
    same = out[tag][0].tobytes() == out["fold"][0].tobytes()
    print(tag, "same" if same else "DIFF", "iters equal", out[tag][1].tolist() == out["fold"][1].tolist())
    if not same:
        d = np.argwhere(out[tag][0].view(np.uint32) != out["fold"][0].view(np.uint32))
        print(" first diffs (pwm, pos, base):", d[:8].tolist(), "n diffs", len(d))
        i = d[0][0]
        print(" pwm", i, "fold", out["fold"][0][i, d[0][1]], tag, out[tag][0][i, d[0][1]], "iters", out["fold"][1][i], out[tag][1][i])
# one iteration only
for tag, scan in (("fold",0),("ahead",2)):
    ctx.test_em_generation(scan)
    out[tag+"1"] = ctx.em(W, pw, counts, bgd, sat, 0.0, 1)
print("one iteration same:", out["fold1"][0].tobytes() == out["ahead1"][0].tobytes())
