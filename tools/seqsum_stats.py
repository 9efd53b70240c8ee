"""Where the serial EM's scan spends a chain's time (developer build with -DPENGK_SEQSUM_STATS):
     PENGK_BUILD_OUT=stats_libs/stats.so PENGK_EXTRA_FLAGS=-DPENGK_SEQSUM_STATS python peng-motif_amd/build.py --force
     PENGK_LIB=$PWD/stats_libs/stats.so python tools/seqsum_stats.py
   Runs the bench's 16-PWM EM batch once and prints blocks, evaluations per block (1 = no binade crossing) and the
   shader-clock cycles per block in the deposit (incl. the wait for the block's loads) and in the evaluation."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import peng_motif_amd as pk

W, n, L = 10, int(os.environ.get("NSEQ", 10_000_000)), 200
ctx = pk.Context(0)
ctx.synth(1, 0, n, L, W)
counts, ltot, bg = ctx.count_bg(True)
ctx.mirror(W, counts)
V = ctx.bg_model(bg, 2)
bgprob, expected, logp, z = ctx.pattern_stats(W, True, 2, 2, V, ltot, counts)
c = counts.to_host()
seeds = np.lexsort((np.arange(c.size), -c.astype(np.int64)))[:int(os.environ.get("NPWM", 16))]
pw = np.full((len(seeds), W, 4), 0.1, np.float32)
for i, x in enumerate(seeds):
    for q in range(W):
        pw[i, q, (int(x) >> (2 * q)) & 3] = 0.7
bg_k = pk.DeviceArray.from_host(ctx, bgprob.to_host()[2])
ctx.set_option("em_fast", 2)
st = (C.c_ulonglong * 12)()
f = pk.lib().pengk_debug_seqsum_stats
f.argtypes = [C.c_void_p, C.c_int]
assert f(st, 1) == 0
ctx.em(W, pw, counts, bg_k, 1e4, 0.0, 10)
assert f(st, 1) == 0
blocks, evals, dep, ev = st[0], st[1], st[2], st[3]
print("blocks %d  evaluations/block %.3f  fetch wave: deposit + next block's loads, cycles/block %.0f  evaluation cycles/block %.0f  (s_memtime ticks)" %
      (blocks, evals / blocks, dep / blocks, ev / blocks))
print("per block (ticks): row read %.0f  additions %.0f/evaluation  prefix %.0f/evaluation  crossing %.0f/crossing" %
      (st[4] / blocks, st[5] / evals, st[6] / evals, st[7] / max(evals - blocks, 1)))
chains = len(seeds) * 4 * W * 10
print("per chain: %.1f blocks, %.1f extra evaluations (binade crossings)" % (blocks / chains, (evals - blocks) / chains))
if st[10]:
    print("chains walked over evaluated blocks: %d; per chain (ticks): total %.0f = fetch + deposit + row read %.0f + fold_block %.0f + rest; blocks not prefetched %.2f" %
          (st[10], st[8] / st[10], st[2] / st[10], st[3] / st[10], st[9] / st[10]))
