"""Soak of the K1 fuzz (tests/test_gpu_parity.py::test_count_fuzz_against_oracle) beyond the 48 seeds of the suite, with
the slice-overflow path forced on a third of the cases (GPU box).  usage: python tests/tools/count_fuzz.py FIRST_SEED SECONDS"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import peng_motif_amd as pk
import test_gpu_parity as tp

seed0, seconds = int(sys.argv[1]), float(sys.argv[2])
ctx = pk.Context(0)
t_end = time.time() + seconds
seed = seed0
while time.time() < t_end:
    cap = int(np.random.default_rng(seed).choice([0, 0, 64, 128]))
    ctx.set_option("key_cap_override", cap)
    try:
        tp.test_count_fuzz_against_oracle(ctx, seed)
    except AssertionError as e:
        print("MISMATCH seed", seed, "cap", cap, e)
        sys.exit(1)
    finally:
        ctx.set_option("key_cap_override", 0)
    seed += 1
print("count fuzz: seeds %d..%d (key_cap_override 0 / 64 / 128), all tables bit-exact" % (seed0, seed - 1))
