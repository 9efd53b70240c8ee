#!/usr/bin/env python3
"""bench.py -- the PEnG-motif hot path on MI355X: k-mer count -> z-score sweep -> EM.

Workload (BASELINE.json configs[2], the HBM-roofline run): synthetic 10M x 200 bp per GPU
(counter-based generator of SURVEY.md 8d, generated on the device, resident in HBM before the
timed region), W=10, both strands, background order 2.  One "step" = one full pass of the hot path
over that batch:

    K1  4^W k-mer count with K1b (background 3-mers) fused  pengk_count_bg
    C1  all-reduce(sum) of {counts, ltot, bg counts}        torch.distributed (RCCL), N > 1 only
        mirror, background model V                          pengk_mirror_counts / pengk_bg_model
    K2+K3 sweep over 4^W patterns (bgprob, expected, log-p, z)   pengk_pattern_stats
    K5  EM: P seed PWMs x 10 iterations over the 4^W table   pengk_em_device (PWMs split over ranks)

Weak scaling: every rank holds its own 10M-sequence shard (sequences [rank*n, (rank+1)*n) of one
global synthetic set).  `value` is whole-job Gbp/s = bases of all ranks / step time; the component
rates of BASELINE.json's metric (z-scores/s, EM evals/s, count Gbp/s) are reported beside it from
HIP-event timings taken inside the timed steps on the stream the kernels run on.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nseq", type=int, default=10_000_000, help="sequences per GPU")
    ap.add_argument("--len", type=int, default=200, dest="L")
    ap.add_argument("--W", type=int, default=10)
    ap.add_argument("--strand", default="BOTH", choices=["BOTH", "PLUS"])
    ap.add_argument("--pwms", type=int, default=16, help="seed PWMs for the EM phase (whole job)")
    ap.add_argument("--em-iters", type=int, default=10)
    ap.add_argument("--k4-patterns", type=int, default=2000,
                    help="K4 probe after the timed steps: degenerate IUPAC patterns of one hill-climb round (0 = off)")
    ap.add_argument("--em-fast", type=int, default=1, help="pengk option em_fast (one reciprocal per k-mer weight)")
    ap.add_argument("--em-stress-pwms", type=int, default=1000,
                    help="BASELINE configs[4]: EM-only stress on this many seed PWMs (split over ranks), timed after the steps; 0 = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--count-impl", type=int, default=0, help="0 auto, 1 direct atomics, 2 partitioned")
    ap.add_argument("--cpu-sample-seqs", type=int, default=6_000_000)
    args = ap.parse_args()

    import torch
    import peng_motif_amd as pk
    from peng_motif_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # PENGK_BENCH_BACKEND=gloo: rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks
        # (ranks share devices, collectives go through the host).  The measured runs use nccl (= RCCL).
        backend = os.environ.get("PENGK_BENCH_BACKEND", "nccl")
        if backend != "nccl":
            local_rank = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
        local_rank = 0
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d != WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    dev = torch.device("cuda", local_rank)

    W, both, L, nseq = args.W, args.strand == "BOTH", args.L, args.nseq
    NP = 4 ** W
    K = 2
    ctx = pk.Context(local_rank)
    ctx_stream = torch.cuda.Stream(device=dev)
    lib = pk.lib()
    pk._check(lib.pengk_set_stream(ctx.h, ctx_stream.cuda_stream))  # kernels + collectives share one stream
    ctx.set_option("count_impl", args.count_impl)
    ctx.set_option("em_fast", args.em_fast)

    with torch.cuda.stream(ctx_stream):
        # ---- resident input: this rank's shard of the global synthetic set -------------------------
        import ctypes as C
        nw, ni = C.c_uint64(), C.c_uint64()
        pk._check(lib.pengk_synth_sizes(nseq, L, W, 0, C.byref(nw), C.byref(ni)))
        words = torch.empty(nw.value, dtype=torch.int64, device=dev)
        items = torch.empty(max(ni.value, 1), dtype=torch.int64, device=dev)
        ctx.synth(1, rank * nseq, nseq, L, W, 0, words, items)
        nwin = L - W + 1
        sharding.check_global_bin_bound(nseq * ((nwin + W - 1) // W), dist)
        counts = torch.empty(NP, dtype=torch.int32, device=dev)           # uint32 bins (bit pattern)
        scal = torch.zeros(85, dtype=torch.int64, device=dev)             # [0:84] bg counts, [84] ltot
        V = torch.empty(84, dtype=torch.float32, device=dev)
        bgprob = torch.empty((K + 1, NP), dtype=torch.float32, device=dev)
        expected = torch.empty(NP, dtype=torch.float32, device=dev)
        logp = torch.empty(NP, dtype=torch.float32, device=dev)
        z = torch.empty(NP, dtype=torch.float32, device=dev)
        # EM seeds: P PWMs for the whole job, split over ranks; seed k-mers are fixed ids (results are
        # not inspected here -- parity is the tests' job -- only the arithmetic volume matters)
        P_total = args.pwms
        my_pwms = [i for i in range(P_total) if i % world == rank]
        n_my = len(my_pwms)
        rng = np.random.default_rng(5)
        seed_ids = rng.integers(0, NP, size=P_total)
        pw0 = np.full((max(n_my, 1), W, 4), 0.1, np.float32)
        for j, i in enumerate(my_pwms):
            for q in range(W):
                pw0[j, q, (int(seed_ids[i]) >> (2 * q)) & 3] = 0.7
        pw_init = torch.from_numpy(pw0).to(dev)
        pwms = torch.empty_like(pw_init)
        em_state = torch.zeros((max(n_my, 1), 2), dtype=torch.int32, device=dev)
        em_change = torch.zeros(max(n_my, 1), dtype=torch.float32, device=dev)
        alpha = np.ones(3, np.float32)

        ev = {k: [ctx.timer(), ctx.timer()] for k in ("count", "sweep", "em")}
        acc = {k: 0.0 for k in ev}
        pending = []

        def step(timed):
            if timed:
                ctx.record(ev["count"][0])
            # K1 with K1b fused into the same scan
            pk._check(lib.pengk_count_bg(ctx.h, int(both), counts.data_ptr(), scal[84:].data_ptr(), scal.data_ptr()))
            if timed:
                ctx.record(ev["count"][1])
            sharding.allreduce_tables(counts, scal, dist)  # the ONE exchange step (C1); no-op at N = 1
            if both:
                pk._check(lib.pengk_mirror_counts(ctx.h, W, counts.data_ptr()))
            pk._check(lib.pengk_bg_model(ctx.h, scal.data_ptr(), K, alpha.ctypes.data, V.data_ptr()))
            if timed:
                ctx.record(ev["sweep"][0])
            pk._check(lib.pengk_pattern_stats(ctx.h, W, int(both), K, K, V.data_ptr(), scal[84:].data_ptr(), counts.data_ptr(),
                                              bgprob.data_ptr(), expected.data_ptr(), logp.data_ptr(), z.data_ptr()))
            if timed:
                ctx.record(ev["sweep"][1])
            pwms.copy_(pw_init)
            if timed:
                ctx.record(ev["em"][0])
            if n_my:
                pk._check(lib.pengk_em_device(ctx.h, W, n_my, pwms.data_ptr(), 1e4, 0.0, args.em_iters, counts.data_ptr(),
                                              bgprob[K].data_ptr(), em_state.data_ptr(), em_change.data_ptr()))
            if timed:
                ctx.record(ev["em"][1])

        def barrier():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        for _ in range(args.warmup):
            step(False)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(True)
            # events are re-recorded every step: read them before the next record (this sync sits inside the
            # timed region on purpose: it is part of what a host driving the path pays per batch)
            for k in ev:
                acc[k] += ctx.elapsed_ms(ev[k][0], ev[k][1])
        barrier()
        dt = time.perf_counter() - t0

        # ---- EM-only stress (BASELINE configs[4]): many seed PWMs on the table the last step left -------------
        em_stress = None
        if args.em_stress_pwms > 0:
            n_st = len([i for i in range(args.em_stress_pwms) if i % world == rank])
            ids = rng.integers(0, NP, size=max(n_st, 1))
            ps0 = np.full((max(n_st, 1), W, 4), 0.1, np.float32)
            for j in range(n_st):
                for q in range(W):
                    ps0[j, q, (int(ids[j]) >> (2 * q)) & 3] = 0.7
            ps_init = torch.from_numpy(ps0).to(dev)
            ps = torch.empty_like(ps_init)
            st_state = torch.zeros((max(n_st, 1), 2), dtype=torch.int32, device=dev)
            st_change = torch.zeros(max(n_st, 1), dtype=torch.float32, device=dev)
            t_a, t_b = ctx.timer(), ctx.timer()
            best = None
            for rep in range(3):
                ps.copy_(ps_init)
                ctx.record(t_a)
                if n_st:
                    pk._check(lib.pengk_em_device(ctx.h, W, n_st, ps.data_ptr(), 1e4, 0.0, args.em_iters, counts.data_ptr(),
                                                  bgprob[K].data_ptr(), st_state.data_ptr(), st_change.data_ptr()))
                ctx.record(t_b)
                ms = ctx.elapsed_ms(t_a, t_b)
                best = ms if best is None else min(best, ms)
            em_stress = (n_st, best)

        # ---- the bit-exact serial EM mode (what the CLI runs) on the step's own PWM batch, for the record ----------------
        em_serial_ms = None
        if n_my and args.em_fast != 2:
            ctx.set_option("em_fast", 2)
            t_a, t_b = ctx.timer(), ctx.timer()
            for rep in range(2):
                pwms.copy_(pw_init)
                ctx.record(t_a)
                pk._check(lib.pengk_em_device(ctx.h, W, n_my, pwms.data_ptr(), 1e4, 0.0, args.em_iters, counts.data_ptr(),
                                              bgprob[K].data_ptr(), em_state.data_ptr(), em_change.data_ptr()))
                ctx.record(t_b)
                ms = ctx.elapsed_ms(t_a, t_b)
                em_serial_ms = ms if em_serial_ms is None else min(em_serial_ms, ms)
            ctx.set_option("em_fast", args.em_fast)

        # ---- K4 probe (not part of the step): the mutants of one hill-climb round, 1 .. 6 degenerate letters each ----
        k4 = None
        if args.k4_patterns > 0:
            rk = np.random.default_rng(7)
            sizes = np.array([1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 4])
            ids, members = [], 0
            for _ in range(args.k4_patterns):
                x = int(rk.integers(0, NP))
                letters = [(x >> (2 * q)) & 3 for q in range(W)]
                for q in rk.choice(W, size=int(rk.integers(1, 7)), replace=False):
                    letters[q] = int(rk.integers(4, 11))
                ids.append(sum(l * 11 ** q for q, l in enumerate(letters)))
                members += int(np.prod(sizes[letters]))
            ids = np.array(ids, dtype=np.uint64)
            st = np.zeros(len(ids) * 24, np.uint8)
            best = None
            for rep in range(3):
                ctx.synchronize()
                t_k = time.perf_counter()
                pk._check(lib.pengk_iupac_aggregate(ctx.h, W, int(both), ids.ctypes.data, len(ids), counts.data_ptr(),
                                                    bgprob[K].data_ptr(), expected.data_ptr(), st.ctypes.data))
                d_k = time.perf_counter() - t_k
                best = d_k if best is None else min(best, d_k)
            k4 = (len(ids), members, best)

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    total_bases = nseq * L * world
    value = total_bases / (dt / args.steps) / 1e9

    if rank == 0:
        ltot = int(scal[84].item())
        count_ms = acc["count"] / args.steps
        sweep_ms = acc["sweep"] / args.steps
        em_ms = acc["em"] / args.steps
        # algorithmic bytes of K1 per launch: packed payload + 8 B per scan item + the count table
        alg_bytes = (nseq * L + 3) // 4 + 8 * int(ni.value) + 4 * NP
        achieved = alg_bytes / (count_ms * 1e-3) / 1e9 if count_ms > 0 else 0.0
        out = {
            "metric": "4^W pattern z-scores/s + EM PWM-kmer evals/s at W=10; Gbp/s k-mer count",
            "value": round(value, 4), "unit": "Gbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 counts / f32 scores / f64 EM accumulators", "data": "synthetic",
            "config": {"workload": "synthetic %dx%d bp per GPU, W=%d, %s strands, bg-order 2 (BASELINE configs[2]); step = count + all-reduce + sweep + EM(%d PWMs x %d it)"
                       % (nseq, L, W, "both" if both else "plus", P_total, args.em_iters),
                       "n_seq_per_gpu": nseq, "seq_len": L, "W": W, "strand": args.strand, "ltot_global": ltot,
                       "parallelism": "sequence shards x%d, one all-reduce of the 4^W counts" % world,
                       "em_mode": {0: "reference terms, fp64 tree sums (1e-5 rel.)", 1: "one reciprocal per weight, fp64 tree sums (1e-5 rel., BASELINE.json's bar)",
                                   2: "serial float32, bit-exact"}[args.em_fast]
                                  + "; the bit-exact serial mode the CLI uses is timed separately: components.em_serial_mode_ms"},
            "components": {
                "count_gbp_per_s_per_gpu": round(nseq * L / (count_ms * 1e-3) / 1e9, 3) if count_ms else None,
                "zscores_per_s": round(NP / (sweep_ms * 1e-3), 1) if sweep_ms else None,
                "em_evals_per_s_per_gpu": round(n_my * args.em_iters * NP / (em_ms * 1e-3), 1) if em_ms and n_my else None,
                "count_ms": round(count_ms, 4), "sweep_ms": round(sweep_ms, 4), "em_ms": round(em_ms, 4),
                "em_stress_pwms_per_gpu": em_stress[0] if em_stress else None,
                "em_stress_ms": round(em_stress[1], 4) if em_stress else None,
                "em_stress_evals_per_s_per_gpu": round(em_stress[0] * args.em_iters * NP / (em_stress[1] * 1e-3), 1)
                if em_stress and em_stress[1] else None,
                "em_serial_mode_ms": round(em_serial_ms, 4) if em_serial_ms else None,
                # K4 (host call incl. id upload, result download and the libm epilogue): patterns and underlying k-mers
                "k4_patterns": k4[0] if k4 else None, "k4_ms": round(k4[2] * 1e3, 4) if k4 else None,
                "k4_kmers_visited_per_s": round(k4[1] / k4[2], 1) if k4 else None,
            },
            "roofline": {"kernel": "pengk_count_bg = count_scatter_kernel<%d,%s> + count_hist_kernel + count_gather_kernel (K1, K1b fused)"
                                   % (W, "both" if both else "plus") if W in (8, 10) and args.count_impl != 1
                                   else "pengk_count_bg = count_scatter12_kernel<%s> + count_rescatter12_kernel + count_hist_kernel + count_gather12_kernel (K1 two-level, K1b fused)"
                                   % ("both" if both else "plus") if W == 12 and args.count_impl != 1
                                   else "pengk_count_bg = count_kernel<%d,%s> (direct atomics)" % (W, "both" if both else "plus"),
                         "bound": "hbm",
                         "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        prof = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(prof):
            try:
                out["roofline"]["traffic"] = json.load(open(prof)).get("count_kernel_hbm_bytes_per_launch")
            except Exception:
                pass
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, W, both, L)
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(args, W, both, L):
    """The oracle (bit-exact restatement of the reference's serial loops) timed on one host core on a
    bounded sample of the same workload: count over `cpu_sample_seqs` sequences, the sweep over all 4^W
    patterns, one EM iteration of 4 PWMs."""
    from oracle import oracle as po
    n = args.cpu_sample_seqs
    codes, offs = po.synth(1, 0, n, L)
    t0 = time.perf_counter()
    counts, ltot = po.count(codes, offs, W, both)
    t_count = time.perf_counter() - t0
    V = po.bg_V(po.bg_counts(codes[: 2000 * L], offs[:2001], 2), 2)
    nsweep = 10
    t0 = time.perf_counter()
    for _ in range(nsweep):
        bgp = [po.bgprob(W, k, V, both) for k in range(3)]
        e, lp, z = po.stats(W, counts, bgp[2], ltot)
    t_sweep = (time.perf_counter() - t0) / nsweep
    pw = np.full((W, 4), 0.1, np.float32)
    pw[:, 0] = 0.7
    t0 = time.perf_counter()
    npw = 30
    for _ in range(npw):
        po.em(W, counts, bgp[2], pw, 1e4, 0.0, 10, mode=0)
    t_em = time.perf_counter() - t0
    extra = reference_probe(W, both, L)
    return {"value": round(n * L / t_count / 1e9, 5), "unit": "Gbp/s", "cores": 1, "kind": "port", **extra,
            "sample": "oracle (bit-exact port of the reference's serial loops, g++ -O3, 1 thread) on %d x %d bp of the same synthetic set: count %.2f s; %d sweeps over 4^%d patterns %.3f s each; %d PWMs x 10 EM iterations %.2f s"
                      % (n, L, t_count, nsweep, W, t_sweep, npw, t_em),
            "zscores_per_s": round(4 ** W / t_sweep, 1), "em_evals_per_s": round(npw * 10 * 4 ** W / t_em, 1)}


def reference_probe(W, both, L, n=400_000):
    """When the build of the real reference travelled with the repository (oracle/_ref/ref_dump, built in the
    build container from /root/reference), time ITS BasePattern constructor (background probabilities + count +
    expected / log-p / z: src/base_pattern.cpp:17-64) on a sample of the same synthetic set, on this host.
    Reported beside the port so that the port can be seen not to flatter the GPU."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
    if not os.path.exists(exe):
        return {}
    import subprocess
    import tempfile
    from oracle import oracle as po
    try:
        tmp = tempfile.mkdtemp(prefix="pengk_refprobe_")
        codes, _ = po.synth(1, 0, n, L)
        rows = np.frombuffer(b"NACGT", dtype=np.uint8)[codes].reshape(n, L)
        with open(os.path.join(tmp, "s.fa"), "wb") as f:
            f.write(b"".join((">s%d\n" % i).encode() + rows[i].tobytes() + b"\n" for i in range(n)))
        subprocess.run([exe, os.path.join(tmp, "s.fa"), str(W), "BOTH" if both else "PLUS", tmp, "tables"], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
        meta = dict(l.split() for l in open(os.path.join(tmp, "meta.txt")))
        t = float(meta["basepattern_seconds"])
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
        return {"reference_basepattern_gbp_per_s": round(n * L / t / 1e9, 5),
                "reference_sample": "compiled reference (oracle/_ref, g++ -O3, 1 thread): BasePattern constructor on %d x %d bp in %.2f s" % (n, L, t)}
    except Exception as e:  # the probe is optional
        return {"reference_sample": "probe failed: %r" % (e,)}


if __name__ == "__main__":
    main()
