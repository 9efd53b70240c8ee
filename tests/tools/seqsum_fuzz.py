"""Fuzz of pengk_sequential_sum_f32 against numpy's left-to-right float32 sum (GPU box).
usage: python tests/tools/seqsum_fuzz.py FIRST_SEED SECONDS
Random chains mixing the regimes the scan distinguishes: uniform terms, the whole exponent range, exact half-ulps of the
running sum (ties), sparse chains with giants, denormals, near-overflow terms; lengths from 1 to 400k."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import peng_motif_amd as pk


def chain(rng, n):
    parts, left = [], n
    while left > 0:
        m = int(min(left, rng.integers(1, max(2, n // 2 + 1))))
        kind = int(rng.integers(0, 7))
        if kind == 0:
            t = rng.random(m, dtype=np.float32) * np.float32(10.0 ** rng.integers(-10, 10))
        elif kind == 1:
            with np.errstate(over="ignore", under="ignore"):
                t = np.ldexp(1.0 + rng.random(m), rng.integers(-149, 100, m)).astype(np.float32)
        elif kind == 2:  # multiples of a half-ulp of some magnitude: ties
            u = np.float32(2.0 ** int(rng.integers(-40, 20)))
            t = (rng.integers(0, 6, m).astype(np.float32) * u).astype(np.float32)
        elif kind == 3:
            t = np.zeros(m, np.float32)
            idx = rng.integers(0, m, max(1, m // 50))
            t[idx] = (rng.random(idx.size) * 10.0 ** rng.integers(-20, 20, idx.size)).astype(np.float32)
        elif kind == 4:
            t = (rng.integers(0, 2 ** int(rng.integers(1, 24)), m).astype(np.float32) * np.float32(1e-45)).astype(np.float32)
        elif kind == 5:
            t = np.full(m, np.float32(2.0 ** int(rng.integers(-30, 30))), np.float32)
        else:
            t = rng.lognormal(-6.0, 4.0, m).astype(np.float32)
        t[~np.isfinite(t)] = 0
        parts.append(t)
        left -= m
    return np.concatenate(parts)[:n]


def main():
    seed0, seconds = int(sys.argv[1]), float(sys.argv[2])
    ctx = pk.Context(0)
    t_end = time.time() + seconds
    seed, n_chains, n_terms = seed0, 0, 0
    while time.time() < t_end:
        rng = np.random.default_rng(seed)
        n = int(rng.choice([1, 63, 64, 4095, 4096, 4097, 8191, 12288, 100000, 262144, 400000]) if rng.random() < 0.5 else rng.integers(1, 400000))
        rows = np.stack([chain(rng, n) for _ in range(int(rng.integers(1, 6)))])
        with np.errstate(over="ignore", invalid="ignore"):
            want = np.cumsum(rows, axis=1, dtype=np.float32)[:, -1]
        got = ctx.sequential_sum(rows)
        if got.view(np.uint32).tolist() != want.view(np.uint32).tolist():
            print("MISMATCH seed", seed, "n", n, got, want)
            np.save("gpurun_out/seqsum_fuzz_fail_%d.npy" % seed, rows)
            sys.exit(1)
        n_chains += rows.shape[0]
        n_terms += rows.size
        seed += 1
    print("seqsum fuzz: seeds %d..%d, %d chains, %.3g terms, all bit-exact" % (seed0, seed - 1, n_chains, n_terms))


main()
