#!/usr/bin/env python3
"""bench.py -- the PEnG-motif hot path on MI355X: k-mer count -> z-score sweep -> EM.

Workload (BASELINE.json configs[2], the HBM-roofline run): synthetic 10M x 200 bp per GPU
(counter-based generator of SURVEY.md 8d, generated on the device, resident in HBM before the
timed region), W=10, both strands, background order 2.  One "step" = one full pass of the hot path
over that batch:

    K1  4^W k-mer count with K1b (background 3-mers) fused  pengk_count_bg
    C1  all-reduce(sum) of {counts, ltot, bg counts}        pengk_allreduce_tables (RCCL over xGMI, N > 1 only)
        mirror, background model V                          pengk_mirror_counts / pengk_bg_model
    K2+K3 sweep over 4^W patterns (bgprob, expected, log-p, z)   pengk_pattern_stats
    K5  EM: P seed PWMs x 10 iterations over the 4^W table   pengk_em_device (PWMs split over ranks)

The EM inside the step runs in the mode the peng_motif CLI ships with (em_fast = 2: the reference's float32 sums in the
reference's order, bit-exact); the library's throughput mode (em_fast = 1, within BASELINE.json's 1e-5) is timed on the
same batch and reported beside it.

Weak scaling: every rank holds its own 10M-sequence shard (sequences [rank*n, (rank+1)*n) of one
global synthetic set).  `value` is whole-job Gbp/s = bases of all ranks / step time; the component
rates of BASELINE.json's metric (z-scores/s, EM evals/s, count Gbp/s) are reported beside it from
HIP-event timings taken inside the timed steps on the stream the kernels run on.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.  Started without a launcher
and N > 1 it starts `python -m torch.distributed.run` itself (before anything touches the GPU) and relays the line.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
FP32_VECTOR_PEAK_TF = 157.3  # dense fp32 vector peak (same guide); K5 has no contraction for the matrix cores


def parse_args():
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
    ap.add_argument("--em-fast", type=int, default=2,
                    help="pengk option em_fast inside the step: 2 = serial bit-exact (the CLI's mode), 1 = one reciprocal per weight, 0 = reference terms")
    ap.add_argument("--em-stress-pwms", type=int, default=1000,
                    help="BASELINE configs[4]: EM-only stress on this many top-count seeds of the PLUS table (split over ranks); 0 = skip")
    ap.add_argument("--em-serial-scan", type=int, default=2,
                    help="the serial EM's scheme (pengk_test_em_generation): 2 = blocks evaluated ahead of the chain, three launches per iteration (default), 3 = two launches, 1 = scan block after block, 0 = fold")
    ap.add_argument("--em-overlap", type=int, default=0,
                    help="pengk option em_overlap: streams the serial EM's batches of PWMs take turns on (1..4; 0 = the library's default)")
    ap.add_argument("--em-table-budget-mb", type=int, default=0,
                    help="pengk option em_table_budget_mb: weight tables per batch of PWMs in the serial EM mode (0 = automatic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end peng_motif CLI run on the config's FASTA")
    ap.add_argument("--e2e-runs", type=int, default=5, help="runs of the peng_motif CLI on the config's FASTA (the median is reported)")
    ap.add_argument("--e2e-pause", type=float, default=1.0, help="seconds between those runs (a finished run's teardown in the driver slows the next one's start)")
    ap.add_argument("--count-impl", type=int, default=0, help="0 auto, 1 direct atomics, 2 partitioned")
    ap.add_argument("--cpu-sample-seqs", type=int, default=6_000_000)
    ap.add_argument("--checks", action="store_true", help="(kept for old command lines: the checksums are always on the line now)")
    ap.add_argument("--pipelined", type=int, default=1, help="1: also time the passes as a two-stream pipeline (components.pipelined); 0: skip")
    ap.add_argument("--config3-steps", type=int, default=3,
                    help="timed steps of the BASELINE configs[3] leg (W=12, one 12.5M x 200 bp shard per rank, 64 MiB exchange); 0 = skip")
    ap.add_argument("--config3-nseq", type=int, default=12_500_000, help="sequences per rank in the configs[3] leg")
    ap.add_argument("--strong", type=int, default=1,
                    help="1: also run the line's sequence set STRONG-scaled (split over the ranks; components.config2_strong); 0: skip")
    return ap.parse_args()


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` without a launcher: start one rank per GPU as a CHILD process (this process has not
    touched the GPU and never execs) and relay its output."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def sha(t):
    return hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest()


def config_key(nseq, L, W, both):
    return "W%d_n%d_L%d_%s" % (W, nseq, L, "BOTH" if both else "PLUS")


def baseline_config_name(nseq, L, W, both):
    """Which BASELINE.json config a size is -- only the sizes BASELINE.json names get its label."""
    if (nseq, L, W, both) == (10_000_000, 200, 10, True):
        return "BASELINE configs[2]"
    if (nseq, L, W, both) == (12_500_000, 200, 12, True):
        return "one of the 8 shards of BASELINE configs[3]"
    if (nseq, L, W, both) == (10_000_000, 200, 10, False):
        return "the PLUS table of BASELINE configs[4]"
    return "not a BASELINE.json configuration"



class Runtime:
    """what a leg needs from main(): modules, the context, the process group, the rank geometry"""


def max_over_ranks(rt, x):
    if not rt.multi:
        return x
    t = rt.torch.tensor([x], dtype=rt.torch.float64, device=rt.dev if rt.dist.get_backend() == "nccl" else "cpu")
    rt.dist.all_reduce(t, op=rt.dist.ReduceOp.MAX)
    return float(t.item())


def sweep_roofline(W, sweep_ms):
    """K2+K3 against the HBM roofline: SURVEY.md 8(d)'s 28 B per pattern at max_k = 2 -- 4 B of count read, (2 + 1) x 4 B of
    background probabilities, expected, log-p and z written -- over the HIP-event time of pengk_pattern_stats."""
    alg = 28 * 4 ** W
    ach = alg / (sweep_ms * 1e-3) / 1e9 if sweep_ms and sweep_ms > 0 else 0.0
    traffic, source = None, None
    try:  # (the PMC passes of the committed profile, like K1's: the driver cannot run rocprofv3)
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic_by_config.json"))).get("sweep_W%d" % W)
        if t:
            traffic, source = t["hbm_bytes_per_launch"], "profiles/traffic_by_config.json[sweep_W%d]: rocprofv3 --pmc passes at commit %s, not this run" % (W, t["commit"])
    except Exception:
        pass
    return {"kernel": "stats_kernel<%d> (K2+K3: background probabilities of orders 0..2, strand aggregation, expected, log-p, z)" % W,
            "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
            "traffic": traffic, "traffic_source": source, "algorithmic_bytes_per_launch": alg, "ms": round(sweep_ms, 5),
            "note": "28 B per pattern (SURVEY.md 8d); at W = 10 the 29 MB stay in L2 / Infinity Cache and the launch is ~20 us: launch-bound, not byte-bound"}


def golden_row(W, nseq, L, both, world):
    """What `world` ranks holding sequences [r * nseq, (r + 1) * nseq) of the seed-1 set must produce together, derived
    from the COMPILED REFERENCE shard by shard (tests/golden/make_shard_golden.py): sha256 of the summed (mirrored) count
    table, ltot, the 84 background counters, and z from the oracle's sweep on those sums.  None if nobody computed it."""
    path = os.path.join(ROOT, "tests", "golden", "shard_prefix_checksums.json")
    if not os.path.exists(path):
        return None
    for r in json.load(open(path)):
        if (r["kind"], r.get("k"), r["W"], r["n_per_shard"], r["L"], r["strand"], r["seed"]) == \
                ("prefix", world, W, nseq, L, "BOTH" if both else "PLUS", 1):
            return r
    return None


def run_leg(rt, W, both, L, nseq, P_total, em_iters, steps, warmup, em_fast, seq0=None, local_only=False, row=None):
    """One configuration through the whole step, `warmup` untimed + `steps` timed passes bracketed by barriers:
    K1(+K1b) count -> C1 exchange -> mirror, V -> K2+K3 sweep -> K5 EM on this rank's share of P_total seed PWMs.
    This rank holds sequences [seq0, seq0 + nseq) of the seed-1 set (default: rank * nseq, the weak-scaling layout).
    local_only: no exchange, all P_total PWMs on this rank -- what ONE rank does with the sequences it is given, run by
    every rank on its own (the denominator of a strong-scaling figure).  row = (n_per_shard, k): the reference-derived
    row the tables are to be compared with (default: this layout's, (nseq, world)).
    Returns the tables, HIP-event times per component, the wall time of the timed steps (max over ranks) and the checks:
    sha256 of what the ranks hold after the exchange, compared with that row."""
    torch, pk, lib, C, ctx, dist = rt.torch, rt.pk, rt.lib, rt.C, rt.ctx, rt.dist
    rank, world, dev = rt.rank, rt.world, rt.dev
    NP, K = 4 ** W, 2
    leg = Runtime()
    exchanging = rt.multi and not local_only
    # ---- resident input: this rank's shard of the global synthetic set -------------------------
    nw, ni = C.c_uint64(), C.c_uint64()
    pk._check(lib.pengk_synth_sizes(nseq, L, W, 0, C.byref(nw), C.byref(ni)))
    words = torch.empty(nw.value, dtype=torch.int64, device=dev)
    items = torch.empty(max(ni.value, 1), dtype=torch.int64, device=dev)
    ctx.synth(1, rank * nseq if seq0 is None else seq0, nseq, L, W, 0, words, items)
    nwin = L - W + 1
    if exchanging and rt.rccl_ranks:
        pk._check(lib.pengk_comm_check_bin_bound(ctx.h))
    elif exchanging:
        rt.sharding.check_global_bin_bound(nseq * ((nwin + W - 1) // W), dist)
    counts = torch.empty(NP, dtype=torch.int32, device=dev)           # uint32 bins (bit pattern)
    scal = torch.zeros(85, dtype=torch.int64, device=dev)             # [0:84] bg counts, [84] ltot
    V = torch.empty(84, dtype=torch.float32, device=dev)
    bgprob = torch.empty((K + 1, NP), dtype=torch.float32, device=dev)
    expected = torch.empty(NP, dtype=torch.float32, device=dev)
    logp = torch.empty(NP, dtype=torch.float32, device=dev)
    z = torch.empty(NP, dtype=torch.float32, device=dev)
    # EM seeds of the step: P PWMs for the whole job, split over ranks; fixed pseudo-random seed k-mers (the EM
    # stress uses the seeds SURVEY.md 8d specifies; parity of both is the tests' job)
    my_pwms = [i for i in range(P_total) if local_only or i % world == rank]
    n_my = len(my_pwms)
    rng = np.random.default_rng(5)
    seed_ids = rng.integers(0, NP, size=P_total)

    def seed_pwms(ids):
        pw = np.full((max(len(ids), 1), W, 4), 0.1, np.float32)
        for j, x in enumerate(ids):
            for q in range(W):
                pw[j, q, (int(x) >> (2 * q)) & 3] = 0.7
        return pw

    pw_init = torch.from_numpy(seed_pwms([seed_ids[i] for i in my_pwms])).to(dev)
    pwms = torch.empty_like(pw_init)
    em_state = torch.zeros((max(n_my, 1), 2), dtype=torch.int32, device=dev)
    em_change = torch.zeros(max(n_my, 1), dtype=torch.float32, device=dev)
    alpha = np.ones(3, np.float32)
    ctx.set_option("em_fast", em_fast)

    ev = {k: [ctx.timer(), ctx.timer()] for k in ("count", "exchange", "sweep", "em")}
    acc = {k: 0.0 for k in ev}

    def exchange():
        if not exchanging:
            return
        if rt.rccl_ranks:  # the ONE exchange step (C1)
            pk._check(lib.pengk_allreduce_tables(ctx.h, W, counts.data_ptr(), scal[84:].data_ptr(), scal.data_ptr()))
        else:               # no-op at N = 1; gloo rehearsal otherwise
            rt.sharding.allreduce_tables(counts, scal, dist)

    def step(timed):
        if timed:
            ctx.record(ev["count"][0])
        # K1 with K1b fused into the same scan
        pk._check(lib.pengk_count_bg(ctx.h, int(both), counts.data_ptr(), scal[84:].data_ptr(), scal.data_ptr()))
        if timed:
            ctx.record(ev["count"][1])
            ctx.record(ev["exchange"][0])
        exchange()
        if timed:
            ctx.record(ev["exchange"][1])
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
            pk._check(lib.pengk_em_device(ctx.h, W, n_my, pwms.data_ptr(), 1e4, 0.0, em_iters, counts.data_ptr(),
                                          bgprob[K].data_ptr(), em_state.data_ptr(), em_change.data_ptr()))
        if timed:
            ctx.record(ev["em"][1])

    def barrier():
        # drain this rank's streams first: the exchange runs on libpengk's communicator, the barrier on torch's, and two
        # communicators must not have collectives in flight on one GPU at the same time
        torch.cuda.synchronize()
        if rt.multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
        # events are re-recorded every step: read them before the next record (this sync sits inside the
        # timed region on purpose: it is part of what a host driving the path pays per batch)
        for k in ev:
            acc[k] += ctx.elapsed_ms(ev[k][0], ev[k][1])
    barrier()
    dt = max_over_ranks(rt, time.perf_counter() - t0)

    # ---- checks (outside the timed region, always): what the ranks hold after the exchange.  `checks` is what a 1-rank
    #      run of the same global set must reproduce bit for bit (serial EM: no batching effects); `checks_ok` compares it
    #      with the reference-derived row for this (W, size, world), if one was computed
    ctx.set_option("em_fast", 2)
    pwms.copy_(pw_init)
    if n_my:
        pk._check(lib.pengk_em_device(ctx.h, W, n_my, pwms.data_ptr(), 1e4, 0.0, em_iters, counts.data_ptr(),
                                      bgprob[K].data_ptr(), em_state.data_ptr(), em_change.data_ptr()))
    ctx.set_option("em_fast", em_fast)
    torch.cuda.synchronize()
    mine = {int(i): pwms[j].cpu().numpy().tobytes().hex() for j, i in enumerate(my_pwms)}
    allp = [mine]
    if exchanging:
        allp = [None] * world
        dist.all_gather_object(allp, mine)
    merged = {}
    for d_ in allp:
        merged.update(d_)
    h_scal = scal.cpu().numpy()
    checks = {"sha_counts": sha(counts.cpu().numpy()), "sha_z": sha(z.cpu().numpy()), "sha_bg_ltot": sha(h_scal),
              "sha_em_pwms": hashlib.sha256("".join(merged[i] for i in sorted(merged)).encode()).hexdigest()}
    row_n, row_k = row if row is not None else (nseq, 1 if local_only else world)
    row = golden_row(W, row_n, L, both, row_k)
    if row is None:
        checks_ok = {"ok": None, "why": "no reference-derived row for W=%d, %d x %d bp per shard, %d shard(s) in tests/golden/shard_prefix_checksums.json" % (W, row_n, L, row_k)}
    else:
        parts = {"counts": checks["sha_counts"] == row["sha_counts_u32"], "ltot": int(h_scal[84]) == row["ltot"],
                 "bg_counters": h_scal[:84].tolist() == row["bgcounts"], "z": checks["sha_z"] == row["sha_z"]}
        checks_ok = {"ok": all(parts.values()), "parts": parts,
                     "against": "tests/golden/shard_prefix_checksums.json[%s, k=%d]: the compiled reference's per-shard tables added (z: the oracle's sweep on those sums)" % (row["job"], row["k"])}

    leg.__dict__.update(dict(counts=counts, scal=scal, V=V, bgprob=bgprob, expected=expected, logp=logp, z=z, pwms=pwms, pw_init=pw_init,
                             em_state=em_state, em_change=em_change, words=words, items=items, ni=ni, n_my=n_my, my_pwms=my_pwms,
                             alpha=alpha, acc=acc, dt=dt, exchange=exchange, step=step, seed_pwms=seed_pwms, checks=checks,
                             checks_ok=checks_ok, barrier=barrier))
    return leg


def pipelined_leg(rt, args, leg, W, both, steps, warmup):
    """The same K passes as the main leg, as a THROUGHPUT pipeline over independent batches (not the line's `value`, which
    stays one pass after the other): the EM of batch i -- 0.85 ms of mostly latency, a few hundred waves -- runs on a second
    stream and context while the first one already counts batch i + 1 (2.9 ms of issue-bound scan on every CU).  Two sets
    of tables; events order count(i) -> EM(i) and EM(i) -> count(i + 2) on the set they share.  Nothing is skipped: every
    pass runs count + exchange + mirror + V + sweep + EM in the CLI's bit-exact mode, and the last two passes' tables and
    PWMs are compared with the main leg's (same input => same bits)."""
    torch, pk, lib, ctx, dist, dev = rt.torch, rt.pk, rt.lib, rt.ctx, rt.dist, rt.dev
    NP, K = 4 ** W, 2
    s0 = torch.cuda.current_stream()
    s1 = torch.cuda.Stream(device=dev)
    ctx2 = pk.Context(dev.index)
    pk._check(lib.pengk_set_stream(ctx2.h, s1.cuda_stream))
    ctx2.set_option("em_fast", args.em_fast)
    ctx2.test_em_generation(args.em_serial_scan)
    sets = []
    for _ in range(2):
        sets.append(dict(counts=torch.empty(NP, dtype=torch.int32, device=dev), scal=torch.zeros(85, dtype=torch.int64, device=dev),
                         V=torch.empty(84, dtype=torch.float32, device=dev), bgprob=torch.empty((K + 1, NP), dtype=torch.float32, device=dev),
                         expected=torch.empty(NP, dtype=torch.float32, device=dev), logp=torch.empty(NP, dtype=torch.float32, device=dev),
                         z=torch.empty(NP, dtype=torch.float32, device=dev), pwms=torch.empty_like(leg.pw_init),
                         state=torch.zeros_like(leg.em_state), change=torch.zeros_like(leg.em_change),
                         ready=torch.cuda.Event(), em_done=torch.cuda.Event()))
    alpha = leg.alpha

    def one(i):
        t = sets[i & 1]
        if i >= 2:
            s0.wait_event(t["em_done"])  # the EM of batch i - 2 has read this set's tables
        pk._check(lib.pengk_count_bg(ctx.h, int(both), t["counts"].data_ptr(), t["scal"][84:].data_ptr(), t["scal"].data_ptr()))
        if rt.rccl_ranks:
            pk._check(lib.pengk_allreduce_tables(ctx.h, W, t["counts"].data_ptr(), t["scal"][84:].data_ptr(), t["scal"].data_ptr()))
        else:
            rt.sharding.allreduce_tables(t["counts"], t["scal"], dist)
        if both:
            pk._check(lib.pengk_mirror_counts(ctx.h, W, t["counts"].data_ptr()))
        pk._check(lib.pengk_bg_model(ctx.h, t["scal"].data_ptr(), K, alpha.ctypes.data, t["V"].data_ptr()))
        pk._check(lib.pengk_pattern_stats(ctx.h, W, int(both), K, K, t["V"].data_ptr(), t["scal"][84:].data_ptr(), t["counts"].data_ptr(),
                                          t["bgprob"].data_ptr(), t["expected"].data_ptr(), t["logp"].data_ptr(), t["z"].data_ptr()))
        t["ready"].record(s0)
        with torch.cuda.stream(s1):
            s1.wait_event(t["ready"])
            t["pwms"].copy_(leg.pw_init)
            if leg.n_my:
                pk._check(lib.pengk_em_device(ctx2.h, W, leg.n_my, t["pwms"].data_ptr(), 1e4, 0.0, args.em_iters, t["counts"].data_ptr(),
                                              t["bgprob"][K].data_ptr(), t["state"].data_ptr(), t["change"].data_ptr()))
            t["em_done"].record(s1)

    for i in range(warmup):
        one(i)
    leg.barrier()
    t0 = time.perf_counter()
    for i in range(warmup, warmup + steps):
        one(i)
    leg.barrier()  # (torch.cuda.synchronize: both streams)
    dt = max_over_ranks(rt, time.perf_counter() - t0)
    same = all(bool(torch.equal(t["counts"], leg.counts)) and bool(torch.equal(t["z"].view(torch.int32), leg.z.view(torch.int32)))
               for t in sets)
    ref = leg.pwms if args.em_fast == 2 else None  # (the main leg's last EM ran in the bit-exact mode when that is the step's mode)
    if ref is not None:
        same = same and all(bool(torch.equal(t["pwms"].view(torch.int32), ref.view(torch.int32))) for t in sets)
    ctx2.close()
    ms = dt / steps * 1e3
    return {"ms_per_step": round(ms, 4), "gbp_per_s": round(args.nseq * args.L * rt.world / (ms * 1e-3) / 1e9, 3), "steps": steps,
            "same_bits_as_the_sequential_passes": same,
            "what": "throughput of the same passes as a two-stream pipeline over independent batches: the EM of batch i beside the count "
                    "of batch i + 1 (second context and stream, two sets of tables); NOT the line's value, which runs one pass after the other"}


def config2_strong_leg(rt, args, main_leg_ms, main_checks_ok):
    """BASELINE configs[2] STRONG-scaled, the way `north_star` words its target ("... on 10M x 200 bp at W=10 on one MI355X
    ... and >= 6x further scaling at 8 GPUs"): the SAME 10M-sequence set split N ways -- rank r holds sequences
    [r * n / N, (r + 1) * n / N) --, the same step (count, ONE all-reduce of the 4^W counts, sweep, EM on this rank's share of
    the PWMs).  After the exchange every rank holds the table of the whole set: `checks_ok` compares it with the k = 1 row
    of the weak layout (the union is the same set).  `ms_1rank`: what one rank takes for the whole set, measured in this
    run (every rank runs it on its own, no exchange; at N = 1 it is the line's own step)."""
    W, both, L, total = args.W, args.strand == "BOTH", args.L, args.nseq
    world, rank = rt.world, rt.rank
    what = "the %d x %d bp set of the line's workload split over %d rank(s): sequences [r n / N, (r + 1) n / N) on rank r" % (total, L, world)
    if not rt.multi:
        return {"scaling": "strong", "n_gpus": 1, "ms_per_step": round(main_leg_ms, 4), "ms_1rank": round(main_leg_ms, 4), "speedup_vs_1rank": 1.0,
                "gbp_per_s": round(total * L / (main_leg_ms * 1e-3) / 1e9, 3), "checks_ok": main_checks_ok, "workload": what,
                "note": "one rank: this IS the line's step (nothing is split, nothing exchanged)"}
    lo, hi = rank * total // world, (rank + 1) * total // world
    n = args.steps
    split = run_leg(rt, W, both, L, hi - lo, args.pwms, args.em_iters, n, args.warmup, args.em_fast, seq0=lo, row=(total, 1))
    ms = split.dt / n * 1e3
    comp = {k: round(split.acc[k] / n, 4) for k in ("count", "exchange", "sweep", "em")}
    sha_split, ok_split = split.checks, split.checks_ok
    split = None
    rt.torch.cuda.empty_cache()
    one = run_leg(rt, W, both, L, total, args.pwms, args.em_iters, n, args.warmup, args.em_fast, seq0=0, local_only=True, row=(total, 1))
    ms1 = one.dt / n * 1e3
    same = {k: sha_split[k] == one.checks[k] for k in sha_split}  # (tables, z, background counters AND the PWMs of the EM)
    ok1 = one.checks_ok
    one = None
    rt.torch.cuda.empty_cache()
    return {"scaling": "strong", "n_gpus": world, "ms_per_step": round(ms, 4), "ms_1rank": round(ms1, 4),
            "speedup_vs_1rank": round(ms1 / ms, 3) if ms > 0 else None, "gbp_per_s": round(total * L / (ms * 1e-3) / 1e9, 3),
            "count_ms": comp["count"], "exchange_ms": comp["exchange"], "sweep_ms": comp["sweep"], "em_ms": comp["em"],
            "exchange_bytes": 4 * 4 ** W + 8 * 85, "checks_ok": ok_split, "checks_ok_1rank": ok1,
            "same_bits_as_1rank": {"ok": all(same.values()), "parts": same}, "workload": what,
            "note": "ms_1rank: every rank runs the whole set on its own (no exchange, all PWMs), the slowest counts"}


def config3_leg(rt, args):
    """BASELINE configs[3] as `north_star` words it: 100M x 200 bp, W = 12, sequences sharded over 8 GPUs, one all-reduce
    of the 4^12 counts (64 MiB) -- here with whatever number of ranks the launcher gave: every rank holds one 12.5M-sequence
    shard (rank r = shard r), so N = 8 IS configs[3] and smaller N are its first N shards (weak scaling of that shard)."""
    W, both, L, nseq = 12, True, 200, args.config3_nseq
    leg = run_leg(rt, W, both, L, nseq, args.pwms, args.em_iters, args.config3_steps, 1, args.em_fast)
    n = args.config3_steps
    ms = leg.dt / n * 1e3
    out = {"workload": "synthetic %d x %d bp per GPU, W=12, both strands (%s): count + all-reduce of 4^12 counts + sweep + EM(%d PWMs x %d it)"
                       % (nseq, L, "BASELINE configs[3]" if (nseq, rt.world) == (12_500_000, 8) else
                          "the first %d of the 8 shards of BASELINE configs[3]" % rt.world if nseq == 12_500_000 else "not a BASELINE.json size",
                          args.pwms, args.em_iters),
           "n_gpus": rt.world, "steps": n, "ms_per_step": round(ms, 4), "gbp_per_s": round(nseq * L * rt.world / (ms * 1e-3) / 1e9, 3),
           "count_ms": round(leg.acc["count"] / n, 4), "exchange_ms": round(leg.acc["exchange"] / n, 4),
           "exchange_bytes": 4 * 4 ** W + 8 * 85, "sweep_ms": round(leg.acc["sweep"] / n, 4), "em_ms": round(leg.acc["em"] / n, 4),
           "ltot_global": int(leg.scal[84].item()), "checks": leg.checks, "checks_ok": leg.checks_ok,
           "scaling": "weak (one 12.5M-sequence shard per rank: N = 8 IS configs[3], smaller N are its first N shards)"}
    out["roofline_sweep"] = sweep_roofline(W, leg.acc["sweep"] / n)
    # the leg's own K1 roofline (same definition as the line's: SURVEY.md 8d algorithmic bytes / the HIP-event time of the count)
    count_ms = leg.acc["count"] / n
    alg = (nseq * L + 3) // 4 + 8 * int(leg.ni.value) + 4 * 4 ** W
    ach = alg / (count_ms * 1e-3) / 1e9 if count_ms > 0 else 0.0
    out["roofline"] = {"kernel": "pengk_count_bg = count_scatter12_kernel<both> + count_rescatter12_kernel + count_hist_kernel + count_gather12_kernel (K1 two-level, K1b fused)",
                       "bound": "hbm", "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 6),
                       "traffic": None, "algorithmic_bytes_per_launch": alg}
    prof = os.path.join(ROOT, "profiles", "traffic_by_config.json")
    try:
        pj = json.load(open(prof)).get(config_key(nseq, L, W, both))
        if pj:
            out["roofline"]["traffic"] = pj.get("count_kernel_hbm_bytes_per_launch")
            out["roofline"]["traffic_source"] = "profiles/traffic_by_config.json[%s]: rocprofv3 --pmc passes at commit %s, not this run" % (
                config_key(nseq, L, W, both), pj.get("commit", "?"))
    except Exception:  # noqa: BLE001 -- no profile of this size: no traffic figure
        pass
    return out


def config4_leg(rt, args, W, seed_pwms, counts, bgprob_k, top):
    """BASELINE configs[4]: "1000 seed PWMs, EM-only stress, 1 vs 8 GPUs pattern-space split".  The 1000 top-count seeds
    are dealt round-robin to the ranks; every rank ALSO runs all of them (the 1-rank equivalent, same tables, same
    kernels) so that one run reports both sides; the pieces come back through pengk_allgather (equal blocks, padded) and
    the gathered serial-mode PWMs must equal the 1-rank run's bit for bit (no batching effects in that mode)."""
    torch, pk, lib, ctx, dist = rt.torch, rt.pk, rt.lib, rt.ctx, rt.dist
    rank, world, dev = rt.rank, rt.world, rt.dev
    P = len(top)
    mine = [int(x) for i, x in enumerate(top) if i % world == rank]
    per = (P + world - 1) // world

    def run(ids, mode, reps):
        n = len(ids)
        init = torch.from_numpy(seed_pwms(ids)).to(dev)
        out = torch.empty_like(init)
        state = torch.zeros((max(n, 1), 2), dtype=torch.int32, device=dev)
        change = torch.zeros(max(n, 1), dtype=torch.float32, device=dev)
        ctx.set_option("em_fast", mode)
        t_a, t_b = ctx.timer(), ctx.timer()
        best = None
        for _ in range(reps):
            out.copy_(init)
            ctx.record(t_a)
            if n:
                pk._check(lib.pengk_em_device(ctx.h, W, n, out.data_ptr(), 1e4, 0.0, args.em_iters, counts.data_ptr(), bgprob_k.data_ptr(),
                                              state.data_ptr(), change.data_ptr()))
            ctx.record(t_b)
            ms = ctx.elapsed_ms(t_a, t_b)
            best = ms if best is None else min(best, ms)
        ctx.set_option("em_fast", args.em_fast)
        return out, best

    fast_out, split_fast = run(mine, 1, 3)
    torch.cuda.synchronize()
    fast_pwm0 = fast_out[0].cpu().numpy() if mine else None
    part, split_serial = run(mine, 2, 1)
    # the pieces back to every rank: equal blocks of `per` PWMs (the last ones padded)
    send = torch.zeros((per, W, 4), dtype=torch.float32, device=dev)
    send[:len(mine)] = part[:len(mine)]
    recv = torch.empty((world * per, W, 4), dtype=torch.float32, device=dev)
    t_a, t_b = ctx.timer(), ctx.timer()
    torch.cuda.synchronize()
    ctx.record(t_a)
    if rt.rccl_ranks or not rt.multi:
        pk._check(lib.pengk_allgather(ctx.h, send.data_ptr(), recv.data_ptr(), send.numel() * 4))
    else:  # gloo rehearsal through the host
        parts = [torch.empty((per, W, 4), dtype=torch.float32) for _ in range(world)]
        dist.all_gather(parts, send.cpu())
        recv.copy_(torch.cat(parts).to(dev))
    ctx.record(t_b)
    gather_ms = ctx.elapsed_ms(t_a, t_b)
    gathered = recv.view(world, per, W, 4).permute(1, 0, 2, 3).reshape(world * per, W, 4)[:P]  # round-robin -> seed order
    _, one_fast = run([int(x) for x in top], 1, 2)
    whole, one_serial = run([int(x) for x in top], 2, 1)
    torch.cuda.synchronize()
    same = bool(torch.equal(gathered.view(torch.int32), whole.view(torch.int32)))
    res = {"pwms": P, "iterations": args.em_iters, "n_gpus": world,
           "ms_1rank_equiv": {"serial": round(one_serial, 4), "throughput": round(one_fast, 4)},
           "ms_split": {"serial": round(max_over_ranks(rt, split_serial), 4), "throughput": round(max_over_ranks(rt, split_fast), 4)},
           "allgather_ms": round(max_over_ranks(rt, gather_ms), 4), "allgather_bytes_per_rank": per * W * 16,
           "split_equals_1rank_bit_for_bit": same,
           "note": "ms_split: the slowest rank's share (PWMs dealt round-robin); ms_1rank_equiv: all PWMs on one rank, same tables, same run"}
    return res, (len(mine), split_fast, split_serial), (mine, fast_pwm0)


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_under_torchrun(args))

    if args.gpus != int(os.environ.get("WORLD_SIZE", "1")):
        sys.exit("bench.py: --gpus %d but the launcher started %s rank(s); start it as `python bench.py --gpus N` or with "
                 "--nproc-per-node equal to --gpus" % (args.gpus, os.environ.get("WORLD_SIZE", "1")))
    # ---- end to end, first: the peng_motif CLI on the config's FASTA, before this process touches the GPU -------------
    e2e_state = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not os.environ.get("PENGK_BENCH_FORCE_COMM") and not args.no_e2e:
        e2e_state = e2e_cli(args, args.W, args.strand == "BOTH", args.L, args.nseq)

    # stdout carries the ONE JSON line and nothing else: librccl announces its version there while communicators are
    # built (torch's and libpengk's), so file descriptor 1 points at stderr for the rest of the run and the line goes to
    # a duplicate of the original
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import peng_motif_amd as pk
    from peng_motif_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit("bench.py: --gpus %d but the launcher started %d rank(s); start it as `python bench.py --gpus N` or with "
                 "--nproc-per-node equal to --gpus" % (args.gpus, world))
    dist = None
    backend = None
    # PENGK_BENCH_FORCE_COMM=1: take the N > 1 code path (process group, RCCL communicator, exchange, gathers) with
    # whatever world size the launcher gave, also 1 -- the only way to run it on a one-GPU test box
    multi = world > 1 or bool(os.environ.get("PENGK_BENCH_FORCE_COMM"))
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # PENGK_BENCH_BACKEND=gloo: rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks
        # (ranks share devices, the exchange goes through the host).  The measured runs use RCCL.
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
    dev = torch.device("cuda", local_rank)

    W, both, L, nseq = args.W, args.strand == "BOTH", args.L, args.nseq
    NP = 4 ** W
    K = 2
    ctx = pk.Context(local_rank)
    ctx_stream = torch.cuda.Stream(device=dev)
    lib = pk.lib()
    pk._check(lib.pengk_set_stream(ctx.h, ctx_stream.cuda_stream))  # kernels + the exchange share one stream
    ctx.set_option("count_impl", args.count_impl)
    ctx.set_option("em_fast", args.em_fast)
    ctx.set_option("em_table_budget_mb", args.em_table_budget_mb)
    ctx.test_em_generation(args.em_serial_scan)
    if args.em_overlap:
        ctx.set_option("em_overlap", args.em_overlap)

    import ctypes as C
    rccl_ranks = 0
    exchange_fallback = None
    if multi and backend == "nccl":
        # the exchange step runs in the C++ library (RCCL on the context's stream); torch.distributed only carries the
        # 128-byte communicator id to the ranks, the barriers around the timed region and the max over ranks
        # Safety net (this path has only ever run with one rank on the build's one-GPU box): if the library's communicator
        # cannot be created on some rank, ALL ranks agree to run the exchange as torch.distributed's RCCL all_reduce of
        # the same device buffers instead, and the JSON line says so.
        box = [None]
        comm_error = None
        try:
            if rank == 0:
                buf = C.create_string_buffer(128)
                pk._check(lib.pengk_comm_unique_id(buf))
                box[0] = buf.raw
        except Exception as e:  # noqa: BLE001 -- reported below, after the ranks have agreed
            comm_error = "pengk_comm_unique_id: %s" % e
        dist.broadcast_object_list(box, src=0)
        if box[0] is not None:
            try:
                if os.environ.get("PENGK_BENCH_BREAK_COMM"):  # test hook: exercise the safety net
                    raise RuntimeError("forced by PENGK_BENCH_BREAK_COMM")
                pk._check(lib.pengk_comm_init(ctx.h, box[0], rank, world))
                r_, w_ = C.c_int(), C.c_int()
                pk._check(lib.pengk_comm_info(ctx.h, C.byref(r_), C.byref(w_)))
                assert (r_.value, w_.value) == (rank, world)
                rccl_ranks = w_.value
            except Exception as e:  # noqa: BLE001
                comm_error = "pengk_comm_init: %s" % e
        elif comm_error is None:
            comm_error = "rank 0 could not create the communicator id"
        agreed = torch.tensor([0 if comm_error else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        if int(agreed.item()) == 0:
            if rccl_ranks:
                lib.pengk_comm_destroy(ctx.h)
            rccl_ranks = 0
            exchange_fallback = comm_error or "another rank could not create the library's communicator"
            sys.stderr.write("bench.py rank %d: exchange falls back to torch.distributed all_reduce (%s)\n" % (rank, exchange_fallback))

    rt = Runtime()
    rt.torch, rt.pk, rt.lib, rt.C, rt.sharding, rt.ctx, rt.dist, rt.dev = torch, pk, lib, C, sharding, ctx, dist, dev
    rt.rank, rt.world, rt.multi, rt.rccl_ranks, rt.args = rank, world, multi, rccl_ranks, args

    with torch.cuda.stream(ctx_stream):
        leg = run_leg(rt, W, both, L, nseq, args.pwms, args.em_iters, args.steps, args.warmup, args.em_fast)
        # (names the rest of main() reads)
        counts, scal, V, bgprob, expected, logp, z = leg.counts, leg.scal, leg.V, leg.bgprob, leg.expected, leg.logp, leg.z
        pwms, pw_init, em_state, em_change = leg.pwms, leg.pw_init, leg.em_state, leg.em_change
        words, items, ni = leg.words, leg.items, leg.ni
        n_my, my_pwms, P_total, alpha = leg.n_my, leg.my_pwms, args.pwms, leg.alpha
        acc, dt, exchange, step, seed_pwms = leg.acc, leg.dt, leg.exchange, leg.step, leg.seed_pwms
        checks, checks_ok = leg.checks, leg.checks_ok
        ltot_global, n_items = int(scal[84].item()), int(ni.value)
        pipelined = pipelined_leg(rt, args, leg, W, both, args.steps, max(2, args.warmup)) if args.pipelined else None

        def time_em(mode, n, init, out, state, change, reps=2):
            if not n:
                return None
            ctx.set_option("em_fast", mode)
            t_a, t_b = ctx.timer(), ctx.timer()
            best = None
            for _ in range(reps):
                out.copy_(init)
                ctx.record(t_a)
                pk._check(lib.pengk_em_device(ctx.h, W, n, out.data_ptr(), 1e4, 0.0, args.em_iters, counts.data_ptr(),
                                              bgprob[K].data_ptr(), state.data_ptr(), change.data_ptr()))
                ctx.record(t_b)
                ms = ctx.elapsed_ms(t_a, t_b)
                best = ms if best is None else min(best, ms)
            ctx.set_option("em_fast", args.em_fast)
            return best

        # the other EM mode on the step's own PWM batch, for the record
        other_mode = 1 if args.em_fast == 2 else 2
        em_other_ms = time_em(other_mode, n_my, pw_init, pwms, em_state, em_change)

        # ---- EM-only stress, BASELINE configs[4] as SURVEY.md 8(d) defines it: PLUS count table of this input, the
        #      highest-count k-mers (ties by ascending id) as seeds, threshold 0, all iterations ----------------------
        em_stress = None
        stress_probe = None
        config4 = None
        if args.em_stress_pwms > 0:
            pk._check(lib.pengk_count_bg(ctx.h, 0, counts.data_ptr(), scal[84:].data_ptr(), scal.data_ptr()))
            exchange()
            pk._check(lib.pengk_bg_model(ctx.h, scal.data_ptr(), K, alpha.ctypes.data, V.data_ptr()))
            pk._check(lib.pengk_pattern_stats(ctx.h, W, 0, K, K, V.data_ptr(), scal[84:].data_ptr(), counts.data_ptr(),
                                              bgprob.data_ptr(), expected.data_ptr(), logp.data_ptr(), z.data_ptr()))
            torch.cuda.synchronize()
            c_host = counts.cpu().numpy().view(np.uint32)
            top = np.lexsort((np.arange(NP), -c_host.astype(np.int64)))[:args.em_stress_pwms]
            config4, em_stress, (mine_s, fast_pwm0) = config4_leg(rt, args, W, seed_pwms, counts, bgprob[K], top)
            if rank == 0 and mine_s:  # handed to the CPU-baseline leg, which checks PWM 0 against the oracle
                stress_probe = dict(counts=c_host.copy(), bg=bgprob[K].cpu().numpy(), seed=mine_s[0], pwm=fast_pwm0)
            # back to the step's tables for the K4 probe
            step(False)
            torch.cuda.synchronize()

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

        # ---- BASELINE configs[3]'s shard per rank (W = 12, 12.5M x 200 bp, 64 MiB exchange): its own leg, its own checks ----
        config3 = None
        main_ms = dt / args.steps * 1e3
        if args.config3_steps > 0 and not (W == 12 and nseq == args.config3_nseq):
            leg = step = exchange = words = items = None  # (the closures hold the W = 10 tables)
            counts = scal = V = bgprob = expected = logp = z = pwms = pw_init = em_state = em_change = None
            torch.cuda.empty_cache()
            config3 = config3_leg(rt, args)
        # ---- the line's sequence set strong-scaled: split over the ranks, one all-reduce, same step ----
        strong = None
        if args.strong:
            leg = step = exchange = words = items = None
            counts = scal = V = bgprob = expected = logp = z = pwms = pw_init = em_state = em_change = None
            torch.cuda.empty_cache()
            strong = config2_strong_leg(rt, args, main_ms, checks_ok)

    ms_per_step = dt / args.steps * 1e3
    total_bases = nseq * L * world
    value = total_bases / (dt / args.steps) / 1e9

    if rank == 0:
        ltot = ltot_global
        count_ms = acc["count"] / args.steps
        exchange_ms = acc["exchange"] / args.steps
        sweep_ms = acc["sweep"] / args.steps
        em_ms = acc["em"] / args.steps
        # algorithmic bytes of K1 per launch: packed payload + 8 B per scan item + the count table
        alg_bytes = (nseq * L + 3) // 4 + 8 * n_items + 4 * NP
        achieved = alg_bytes / (count_ms * 1e-3) / 1e9 if count_ms > 0 else 0.0
        mode_name = {0: "reference terms, fp64 tree sums (1e-5 rel.)", 1: "one reciprocal per weight, fp64 tree sums (1e-5 rel., BASELINE.json's bar)",
                     2: "serial float32 in the reference's order, bit-exact (the peng_motif CLI's mode); every cell's chain of roundings evaluated as a wave-wide scan (csrc/seqsum.h)"}
        step_other_ms = (ms_per_step - em_ms + em_other_ms) if em_other_ms is not None else None
        out = {
            "metric": "4^W pattern z-scores/s + EM PWM-kmer evals/s at W=10; Gbp/s k-mer count",
            "value": round(value, 4), "unit": "Gbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            # per-GPU work is fixed as N grows (every rank holds its own 10M-sequence shard): the legs that split a FIXED
            # job over the ranks say so themselves (components.config2_strong, components.config4: "scaling": "strong")
            "scaling": "weak", "vs_baseline": None,
            # how the exchange step ran: ranks of the library's RCCL communicator (0 = no communicator: one rank, or the gloo
            # rehearsal), and -- if the library's communicator could not be created and the ranks agreed to run the exchange
            # as torch.distributed's all_reduce instead -- why.  A measured multi-GPU line has rccl_ranks = n_gpus and null here.
            "rccl_ranks": rccl_ranks, "exchange_fallback": exchange_fallback,
            "dtype": "u32 counts / f32 scores / f32 serial EM sums (f64 tree sums in the throughput EM mode)", "data": "synthetic",
            "config": {"workload": "synthetic %dx%d bp per GPU, W=%d, %s strands, bg-order 2 (%s); step = count + all-reduce + sweep + EM(%d PWMs x %d it)"
                       % (nseq, L, W, "both" if both else "plus", baseline_config_name(nseq, L, W, both), P_total, args.em_iters),
                       "n_seq_per_gpu": nseq, "seq_len": L, "W": W, "strand": args.strand, "ltot_global": ltot,
                       "parallelism": "sequence shards x%d, one all-reduce of the 4^W counts" % world,
                       "exchange": ("RCCL, %d rank(s), pengk_allreduce_tables on the kernels' stream" % rccl_ranks) if rccl_ranks
                       else ("none (one rank)" if not multi else
                             ("RCCL via torch.distributed all_reduce (fallback: %s)" % exchange_fallback) if exchange_fallback else
                             "gloo rehearsal through the host (not a measured configuration)"),
                       "em_mode": mode_name[args.em_fast]},
            "components": {
                "count_gbp_per_s_per_gpu": round(nseq * L / (count_ms * 1e-3) / 1e9, 3) if count_ms else None,
                "zscores_per_s": round(NP / (sweep_ms * 1e-3), 1) if sweep_ms else None,
                "em_evals_per_s_per_gpu": round(n_my * args.em_iters * NP / (em_ms * 1e-3), 1) if em_ms and n_my else None,
                "count_ms": round(count_ms, 4), "exchange_ms": round(exchange_ms, 4), "sweep_ms": round(sweep_ms, 4), "em_ms": round(em_ms, 4),
                "exchange_bytes": 4 * NP + 8 * 85,
                # the step's EM batch in the other mode (1 = throughput mode when the step runs the serial one), and
                # what the step would take with it
                "em_other_mode": other_mode, "em_other_mode_ms": round(em_other_ms, 4) if em_other_ms is not None else None,
                "step_ms_with_other_em_mode": round(step_other_ms, 4) if step_other_ms is not None else None,
                "value_with_other_em_mode": round(total_bases / (step_other_ms * 1e-3) / 1e9, 4) if step_other_ms else None,
                # BASELINE configs[4]: top-count seeds of the PLUS table, threshold 0
                "em_stress_pwms_per_gpu": em_stress[0] if em_stress else None,
                "em_stress_ms": round(em_stress[1], 4) if em_stress and em_stress[1] else None,
                "em_stress_evals_per_s_per_gpu": round(em_stress[0] * args.em_iters * NP / (em_stress[1] * 1e-3), 1)
                if em_stress and em_stress[1] else None,
                "em_stress_serial_mode_ms": round(em_stress[2], 4) if em_stress and em_stress[2] else None,
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
        # HBM traffic and issue counters are NOT measured in this run (they need rocprofv3 --pmc passes): copied from the
        # committed profile OF THIS CONFIGURATION (profiles/traffic_by_config.json, one entry per W / size / strand mode,
        # stamped with its commit); a configuration nobody profiled carries no traffic figure
        prof = os.path.join(ROOT, "profiles", "traffic_by_config.json")
        if os.path.exists(prof):
            try:
                pj = json.load(open(prof)).get(config_key(nseq, L, W, both))
                if pj:
                    out["roofline"]["traffic"] = pj.get("count_kernel_hbm_bytes_per_launch")
                    out["roofline"]["traffic_source"] = "profiles/traffic_by_config.json[%s]: rocprofv3 --pmc passes at commit %s, not this run" % (
                        config_key(nseq, L, W, both), pj.get("commit", "?"))
                    if "issue" in pj:  # second roofline of K1: the pass-A kernel is bound by instruction issue, not by bytes
                        out["roofline_issue"] = pj["issue"]
                else:
                    out["roofline"]["traffic_source"] = "no committed --pmc profile for %s" % config_key(nseq, L, W, both)
            except Exception:
                pass
        if em_stress and em_stress[1]:
            evals = em_stress[0] * args.em_iters * NP / (em_stress[1] * 1e-3)
            out["roofline_em"] = {
                "kernel": "em_accumulate_kernel<%d,...> (K5, throughput mode, %d PWMs x %d iterations)" % (W, em_stress[0], args.em_iters),
                "bound": "fp32 vector issue", "flop_per_eval": 2 * W + 4, "bytes_per_eval": 8,
                "achieved": round(evals * (2 * W + 4) / 1e12, 3), "peak": FP32_VECTOR_PEAK_TF, "unit": "TFLOP/s",
                "frac": round(evals * (2 * W + 4) / 1e12 / FP32_VECTOR_PEAK_TF, 5),
                "table_stream_gb_per_s": round(evals * 8 / 1e9, 1), "table_stream_frac_of_hbm_peak": round(evals * 8 / 1e9 / HBM_PEAK_GBS, 4),
                "note": "8 B per evaluation (u32 count + f32 background) streamed per PWM from L2 / Infinity Cache (the 8 MB of tables stay on die), so the byte rate may exceed the HBM peak; 2W+4 flop per evaluation as SURVEY.md 8(d) counts them"}
            if em_stress[2]:
                # the mode the CLI ships and the bench step times: pinned to the reference's PWMs bit for bit
                ev_s = em_stress[0] * args.em_iters * NP / (em_stress[2] * 1e-3)
                out["roofline_em"]["parity_mode"] = {
                    "kernel": ("em_weights_span_kernel<%d> + em_span_eval_kernel<%d> + em_chain_store_kernel<%d>" % (W, W, W)
                               if W >= 10 and args.em_serial_scan == 2 else
                               "em_span_fused_kernel<%d> + em_chain_store_kernel<%d>" % (W, W) if W >= 10 and args.em_serial_scan == 3 else
                               "em_weights_kernel<%d> + em_fold_scan_kernel<%d>" % (W, W)) + " (K5, serial bit-exact mode, same PWMs)",
                    "ms": round(em_stress[2], 4), "evals_per_s": round(ev_s, 1),
                    "achieved": round(ev_s * (2 * W + 4) / 1e12, 3), "peak": FP32_VECTOR_PEAK_TF, "unit": "TFLOP/s",
                    "frac": round(ev_s * (2 * W + 4) / 1e12 / FP32_VECTOR_PEAK_TF, 5),
                    "note": "reference-parity mode: every cell's 4^(W-1) float32 additions in the reference's order (as a wave-wide scan whose blocks are evaluated ahead of the chain, csrc/seqsum.h); the throughput mode above is pinned to the fp64 oracle within 1e-5, not to the reference's own rounding"}
        # what the ranks hold after the exchange (sha256) and whether that is what the compiled reference's per-shard
        # tables add up to for this (W, size, N) -- at every N, so that a multi-GPU line verifies itself
        out["checks"] = checks
        out["checks_ok"] = checks_ok
        if pipelined is not None:
            out["components"]["pipelined"] = pipelined
        if config3 is not None:
            out["components"]["config3"] = config3
        if config4 is not None:
            config4["scaling"] = "strong (the same 1000 PWMs dealt to the ranks; ms_1rank_equiv is measured in the same run)"
            out["components"]["config4"] = config4
        if strong is not None:
            out["components"]["config2_strong"] = strong
        out["roofline_sweep"] = sweep_roofline(W, sweep_ms)
        if k4:
            # K4 against the HBM roofline: SURVEY.md 8(d)'s 16 B per visited k-mer (8 B count + 4 B expected + 4 B background
            # probability; the device keeps 32-bit counts: 12 B move) over the wall time of one pengk_iupac_aggregate call
            # (id upload, kernels, result download, libm epilogue): a latency-bound batch, nowhere near the byte roofline
            k4_ach = 16 * k4[1] / k4[2] / 1e9
            out["roofline_k4"] = {"kernel": "pengk_iupac_aggregate: iupac_block_kernel / list pipeline + iupac_fold_kernel (K4), %d patterns of one hill-climb round, %d underlying k-mers" % (k4[0], k4[1]),
                                  "bound": "hbm", "achieved": round(k4_ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(k4_ach / HBM_PEAK_GBS, 6),
                                  "traffic": None, "algorithmic_bytes_per_call": 16 * k4[1], "ms": round(k4[2] * 1e3, 4),
                                  "note": "16 B per visited k-mer (SURVEY.md 8d) over the HOST-side wall time of the call, transfers and epilogue included; bound by the latency of a ~0.15 ms call, not by bytes"}
        # everything this process holds on the GPU goes before the host-side legs run: the end-to-end CLI below is its
        # own process on the same card, and its context start-up / exit were measured 0.2 s slower beside a parent that
        # still held 8 GB of device memory and two live contexts
        ctx.close()
        ctx = None
        words = items = counts = scal = V = bgprob = expected = logp = z = pwms = pw_init = leg = None
        torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, W, both, L, stress_probe)
        if e2e_state is not None:
            out["components"]["e2e_cli"] = e2e_reference(e2e_state[0], e2e_state[1], args, W, both, L, nseq, out.get("cpu_baseline"))
            ref_full = out["components"]["e2e_cli"].get("reference_same_box")
            if ref_full is not None and "cpu_baseline" in out:
                out["cpu_baseline"]["reference_full_size"] = ref_full
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    if ctx is not None:
        ctx.close()
    if multi:
        dist.destroy_process_group()


def mem_available_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def reference_same_box(fa, tmp, W, both, L, nseq, our_wall, cpu_base, our_best=None, our_cold=None):
    """The compiled reference CLI (oracle/_ref/peng_motif_ref, its default --threads 1) on the SAME full-size FASTA, on this
    box's host cores, in this run: the denominator north_star asks for.  Skipped -- with the reason -- when the binary did
    not travel, when the box has too little free memory (the reference keeps ~12 bytes per base) or when the
    300k-sequence sample of cpu_baseline extrapolates to more than 300 s."""
    exe = os.path.join(ROOT, "oracle", "_ref", "peng_motif_ref")
    if not os.path.exists(exe):
        return {"skipped": "oracle/_ref/peng_motif_ref not present on this box"}
    need_gb = max(48.0, nseq * L * 14 / 1e9)
    if mem_available_gb() < need_gb:
        return {"skipped": "%.0f GB of host memory available, the reference needs about %.0f GB at this size" % (mem_available_gb(), need_gb)}
    est = None
    try:
        ph = cpu_base["reference_cli_phases"]["threads_1"]
        n_s = int(cpu_base["reference_cli_sample"].split(" on ")[1].split(" x ")[0])
        est = ph["total"] * nseq / n_s
    except Exception:  # noqa: BLE001 -- no sample: run with the time limit alone
        pass
    if est is not None and est > 300.0:
        return {"skipped": "the sample extrapolates to %.0f s (> 300 s)" % est, "extrapolated_s": round(est, 1)}
    meme = os.path.join(tmp, "ref.meme")
    cmd = [exe, fa, "-w", str(W), "--strand", "BOTH" if both else "PLUS", "--threads", "1", "-o", meme]
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    except subprocess.TimeoutExpired:
        return {"skipped": "the reference did not finish within 600 s", "extrapolated_s": round(est, 1) if est else None}
    wall = time.perf_counter() - t0
    if r.returncode != 0:
        return {"skipped": "the reference exited with %d" % r.returncode}
    same = None
    ours = os.path.join(tmp, "o.meme")
    if os.path.exists(ours) and os.path.exists(meme):
        same = open(ours, "rb").read() == open(meme, "rb").read()
    return {"reference_wall_s_same_box": round(wall, 2), "speedup_same_box": round(wall / our_wall, 1),
            "speedup_cold": round(wall / our_cold, 1) if our_cold else None,
            "speedup_policy": "ONE run of the reference (it takes ~50 s) over the MEDIAN of peng_motif's runs (warm: the runs behind the first); the ratio to the FIRST run (speedup_cold) and to the best run are beside it",
            "speedup_same_box_vs_best_run": round(wall / our_best, 1) if our_best else None, "threads": 1,
            "host_cores": os.cpu_count(), "extrapolated_from_sample_s": round(est, 1) if est else None,
            "meme_identical_to_reference": same,
            "command": "oracle/_ref/peng_motif_ref s.fa -w %d --strand %s --threads 1 -o ref.meme (same file, same box, same run)" % (W, "BOTH" if both else "PLUS")}


def e2e_cli(args, W, both, L, nseq):
    """End to end: the config's FASTA on disk -> peng-motif_amd/host/peng_motif (PENGK_TIMING=1) -> MEME + JSON, wall clock
    of the process and the phases it reports, `--e2e-runs` times.  Runs FIRST, before this process has touched the GPU:
    the program is timed the way a user starts it, not beside a parent that holds a HIP context of its own (measured: the
    child's runtime start and exit take ~0.2 s longer then).  Returns (result, scratch directory); the directory keeps
    the FASTA and the MEME file for e2e_reference, which runs the compiled reference on the same file at the end.
    BASELINE.md's 277.3 s (10M x 200 bp, W=10, 1 thread) was measured in the survey container, on another machine: it is
    kept as context and labelled so."""
    exe = os.path.join(ROOT, "peng-motif_amd", "host", "peng_motif")
    gen = os.path.join(ROOT, "tools", "synth_fasta")
    if not (os.path.exists(exe) and os.path.exists(gen)):
        return {"error": "peng_motif / synth_fasta not built"}, None
    import atexit
    import shutil
    import tempfile
    base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 4 * nseq * (L + 12) else None
    tmp = tempfile.mkdtemp(prefix="pengk_e2e_", dir=base)
    atexit.register(shutil.rmtree, tmp, ignore_errors=True)
    if True:  # (the scratch directory outlives this function: e2e_reference / atexit remove it)
        fa = os.path.join(tmp, "s.fa")
        t0 = time.perf_counter()
        subprocess.run([gen, fa, str(nseq), str(L), "1", "0"], check=True, timeout=600, stdout=subprocess.DEVNULL)
        t_gen = time.perf_counter() - t0
        cmd = [exe, fa, "-w", str(W), "--strand", "BOTH" if both else "PLUS", "-o", os.path.join(tmp, "o.meme"), "-j",
               os.path.join(tmp, "o.json")]
        # A freshly leased box can take a while to answer normally: on one of them the first FOUR processes that touched the
        # GPU waited 1.1-2.7 s for their first stream (the fifth: 18 ms; profiles/r04_e2e_experiments.log).  That is the
        # lease waking up, not this program: before the timed runs a tiny input is run, untimed, until the runtime's
        # "first stream" lap is normal twice in a row (twelve tries at most; every lap is reported).
        probe = []
        small = os.path.join(ROOT, "tests", "golden", "MafK_100seqs.fasta")
        if os.path.exists(small):
            for _ in range(12):
                r = subprocess.run([exe, small, "-w", "8", "-o", os.path.join(tmp, "probe.meme")], stdout=subprocess.DEVNULL,
                                   stderr=subprocess.PIPE, env=dict(os.environ, PENGK_TIMING_CREATE="1"), timeout=300)
                if r.returncode != 0:  # (a probe that failed is an error of the leg, never a lap to be tried again)
                    return {"error": "the device-ready probe (peng_motif on MafK_100seqs.fasta) exited %d after %d earlier probe(s): %s"
                                     % (r.returncode, len(probe), r.stderr.decode(errors="replace")[-300:]),
                            "device_ready_probe_first_stream_ms": probe}, None
                lap = [float(l.rsplit(": ", 1)[1].split()[0]) for l in r.stderr.decode(errors="replace").split("\n")
                       if l.startswith("[pengk_create] stream")]
                probe.append(lap[0] if lap else None)
                if len(probe) >= 2 and all(x is not None and x < 100.0 for x in probe[-2:]):
                    break
                time.sleep(0.5)
        runs = []
        for rep in range(args.e2e_runs):  # (the first run also warms the page cache and the GPU code-object cache)
            # A process that has used the GPU is still being torn down in the driver for a while after it has gone, and a
            # process that starts in that window pays for it: back to back the runs took 0.19-0.48 s on one box (runtime
            # start up to 220 ms, exits of 0.15-0.18 s), one second apart 0.185-0.23 s, every one of them
            # (tools/e2e_gap.sh, profiles/r04_e2e_experiments.log).  A user starts the program once: the runs are spaced.
            time.sleep(args.e2e_pause)
            t0 = time.perf_counter()
            p = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                                 env=dict(os.environ, PENGK_TIMING="1", PENGK_TIMING_CREATE="1"))
            phases, create, t_total_line, tail = {}, {}, None, []
            for raw in p.stderr:
                line = raw.decode(errors="replace").rstrip("\n")
                tail.append(line)
                if line.startswith("[timing] ") and ": " in line:
                    k_, v_ = line[len("[timing] "):].rsplit(": ", 1)
                    phases[k_.strip()] = float(v_.split()[0])
                    if k_.strip() == "total":
                        t_total_line = time.perf_counter() - t0
                elif line.startswith("[pengk_create] ") and ": " in line:
                    k_, v_ = line[len("[pengk_create] "):].rsplit(": ", 1)
                    create[k_.strip()] = float(v_.split()[0])
            rc = p.wait(timeout=900)
            wall = time.perf_counter() - t0
            if rc != 0:
                return {"error": "peng_motif exited %d: %s" % (rc, "\n".join(tail)[-300:])}, None
            runs.append({"wall_s": round(wall, 3), "phases_s": phases, "runtime_start_ms": create,
                         # from the program's own "[timing] total" line to the process being gone (what the caller still waits for)
                         "exit_s": round(wall - t_total_line, 3) if t_total_line is not None else None,
                         # from the launch to main()'s clock start + the total main() reports = everything before the exit
                         "before_main_s": round(t_total_line - phases["total"], 3) if t_total_line is not None and "total" in phases else None})
        walls = sorted(r_["wall_s"] for r_ in runs)
        median = walls[len(walls) // 2] if len(walls) % 2 else 0.5 * (walls[len(walls) // 2 - 1] + walls[len(walls) // 2])
        mid = min(runs, key=lambda r_: abs(r_["wall_s"] - median))
        n_motifs = sum(1 for l in open(os.path.join(tmp, "o.meme")) if l.startswith("MOTIF"))
        is_c2 = (nseq, L, W, both) == (10_000_000, 200, 10, True)
        res = {"wall_s": round(median, 3), "wall_s_is": "median of %d runs, %.1f s apart" % (len(runs), args.e2e_pause), "walls_s": [r_["wall_s"] for r_ in runs],
               # the FIRST timed run: page cache of the FASTA warm (it was just written), GPU code objects not yet cached by
               # the driver for this program -- what a user's first start costs
               "cold_first_run_s": runs[0]["wall_s"],
               # where the FASTA lies: "tmpfs" = /dev/shm (memory: no disk in the read), "disk" = the temporary directory's file system
               "fasta_on": "tmpfs" if base else "disk",
               "best_wall_s": walls[0], "phases_s": mid["phases_s"], "runtime_start_ms": mid["runtime_start_ms"],
               "exit_s": mid["exit_s"], "before_main_s": mid["before_main_s"], "runs": runs,
               "device_ready_probe_first_stream_ms": probe,
               "motifs": n_motifs, "fasta_bytes": os.path.getsize(fa),
               "fasta_generation_s": round(t_gen, 2), "command": "peng_motif s.fa -w %d --strand %s -o o.meme -j o.json" % (W, "BOTH" if both else "PLUS"),
               # context only: a different machine (the survey container), NOT a same-box ratio
               "reference_wall_s_baseline_md_other_machine": 277.3 if is_c2 else None,
               "cross_machine_ratio_vs_baseline_md": round(277.3 / median, 1) if is_c2 else None}
        res["measured"] = "before this process touched the GPU (first leg of the run)"
        return res, tmp


def e2e_reference(res, tmp, args, W, both, L, nseq, cpu_base):
    """The compiled reference on the file e2e_cli timed peng_motif on, same box, same run (reference_same_box)."""
    import shutil
    try:
        if tmp and "error" not in res and not args.no_cpu_baseline:
            res["reference_same_box"] = reference_same_box(os.path.join(tmp, "s.fa"), tmp, W, both, L, nseq, res["wall_s"], cpu_base,
                                                           res["best_wall_s"], res.get("cold_first_run_s"))
    finally:
        if tmp:
            shutil.rmtree(tmp, ignore_errors=True)
    return res


def cpu_baseline(args, W, both, L, stress_probe=None):
    """CPU side of the same run, on this box's host cores, on bounded samples of the same workload:
      * the oracle (bit-exact restatement of the reference's serial loops, g++ -O3, one thread): count, bg counts,
        bgprob tables, sweep, seed selection, EM;
      * the compiled reference itself (oracle/_ref, when it travelled with the repository): its BasePattern constructor
        on a sample, and the whole CLI on a sample with every phase of SURVEY.md 8(d) taken from the time its stdout
        banners arrive -- once with 1 thread, once with all cores (count and EM are serial in the reference)."""
    from oracle import oracle as po
    n = args.cpu_sample_seqs
    t0 = time.perf_counter()
    codes, offs = po.synth(1, 0, n, L)
    t0 = time.perf_counter()
    counts, ltot = po.count(codes, offs, W, both)
    t_count = time.perf_counter() - t0
    n_bg = min(n, 500_000)  # the restatement of Sequence.cpp's k-mer arrays is slow: a smaller sample
    t0 = time.perf_counter()
    nb = po.bg_counts(codes[: n_bg * L], offs[: n_bg + 1], 2)
    t_bgc = time.perf_counter() - t0
    V = po.bg_V(nb, 2)
    t0 = time.perf_counter()
    bgp = [po.bgprob(W, k, V, both) for k in range(3)]
    t_bgp = time.perf_counter() - t0
    nsweep = 5
    t0 = time.perf_counter()
    for _ in range(nsweep):
        e, lp, z = po.stats(W, counts, bgp[2], ltot)
    t_stats = (time.perf_counter() - t0) / nsweep
    # the same two sweeps with all cores: the loops the reference runs under OpenMP (src/base_pattern.cpp:232,253,261,289);
    # count and EM are serial in the reference whatever --threads says
    # (the cores this process may run on -- a one-GPU lease is a share of the host --, not the host's hardware threads)
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    po.set_threads(ncores)
    try:
        po.stats(W, counts, bgp[2], ltot)
        t0 = time.perf_counter()
        for _ in range(nsweep):
            po.stats(W, counts, bgp[2], ltot)
        t_stats_all = (time.perf_counter() - t0) / nsweep
        t0 = time.perf_counter()
        for k in range(3):
            po.bgprob(W, k, V, both)
        t_bgp_all = time.perf_counter() - t0
    finally:
        po.set_threads(1)
    t0 = time.perf_counter()
    seeds = po.select(W, z, counts, 10.0, 3, not both, True)
    t_select = time.perf_counter() - t0
    pw = np.full((W, 4), 0.1, np.float32)
    pw[:, 0] = 0.7
    t0 = time.perf_counter()
    npw = 20
    for _ in range(npw):
        po.em(W, counts, bgp[2], pw, 1e4, 0.0, 10, mode=0)
    t_em = time.perf_counter() - t0
    out = {"value": round(n * L / t_count / 1e9, 5), "unit": "Gbp/s", "cores": 1, "kind": "port",
           "sample": "oracle (bit-exact port of the reference's serial loops, g++ -O3, 1 thread) on %d x %d bp of the same synthetic set: count %.2f s; sweep over 4^%d patterns %.3f s; %d PWMs x 10 EM iterations %.2f s"
                     % (n, L, t_count, W, t_stats, npw, t_em),
           "zscores_per_s": round(4 ** W / t_stats, 1), "em_evals_per_s": round(npw * 10 * 4 ** W / t_em, 1),
           # the OpenMP-equivalent sweeps with --threads=all (the port's loops under OpenMP, same values)
           "sweeps_all_threads": {"threads": ncores, "host_hardware_threads": os.cpu_count(), "stats_sweep_s": round(t_stats_all, 5), "zscores_per_s": round(4 ** W / t_stats_all, 1),
                                  "bgprob_tables_3_orders_s": round(t_bgp_all, 4),
                                  "zscores_per_s_incl_bgprob_tables": round(4 ** W / (t_stats_all + t_bgp_all), 1),
                                  "one_thread_zscores_per_s_incl_bgprob_tables": round(4 ** W / (t_stats + t_bgp), 1)},
           "port_phases_s": {"count": round(t_count, 3), "bg_counts_%d_sequences" % n_bg: round(t_bgc, 3), "bgprob_tables_3_orders": round(t_bgp, 3),
                             "stats_sweep": round(t_stats, 4), "seed_selection_%d_seeds" % len(seeds): round(t_select, 3),
                             "em_per_pwm_10_iterations": round(t_em / npw, 4), "sample_sequences": n}}
    if stress_probe is not None:  # BASELINE configs[4] spot check: PWM 0 of the stress batch against the fp64 restatement
        ref, it, _ = po.em(W, stress_probe["counts"].astype(np.uint64), stress_probe["bg"],
                           _seed_pwm(stress_probe["seed"], W), 1e4, 0.0, 10, mode=1, final_norm=False)
        dev = float(np.abs(stress_probe["pwm"].astype(np.float64) - ref).max())
        out["em_stress_check"] = {"seed": po.kmer_str(stress_probe["seed"], W), "max_abs_dev_vs_fp64_oracle": dev, "ok": bool(dev <= 1e-6)}
    out.update(reference_probe(W, both, L))
    out.update(reference_cli_phases(W, both, L))
    return out


def _seed_pwm(x, W):
    pw = np.full((W, 4), 0.1, np.float32)
    for q in range(W):
        pw[q, (int(x) >> (2 * q)) & 3] = 0.7
    return pw


def _write_sample_fasta(path, n, L):
    gen = os.path.join(ROOT, "tools", "synth_fasta")
    if os.path.exists(gen):
        subprocess.run([gen, path, str(n), str(L), "1", "0"], check=True, timeout=600)
        return
    from oracle import oracle as po
    codes, _ = po.synth(1, 0, n, L)
    rows = np.frombuffer(b"NACGT", dtype=np.uint8)[codes].reshape(n, L)
    with open(path, "wb") as f:
        f.write(b"".join((">s%d\n" % i).encode() + rows[i].tobytes() + b"\n" for i in range(n)))


def reference_probe(W, both, L, n=400_000):
    """When the build of the real reference travelled with the repository (oracle/_ref/ref_dump, built in the
    build container from /root/reference), time ITS BasePattern constructor (background probabilities + count +
    expected / log-p / z: src/base_pattern.cpp:17-64) on a sample of the same synthetic set, on this host.
    Reported beside the port so that the port can be seen not to flatter the GPU."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
    if not os.path.exists(exe):
        return {}
    import shutil
    import tempfile
    tmp = tempfile.mkdtemp(prefix="pengk_refprobe_")
    try:
        _write_sample_fasta(os.path.join(tmp, "s.fa"), n, L)
        subprocess.run([exe, os.path.join(tmp, "s.fa"), str(W), "BOTH" if both else "PLUS", tmp, "tables"], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
        meta = dict(l.split() for l in open(os.path.join(tmp, "meta.txt")))
        t = float(meta["basepattern_seconds"])
        return {"reference_basepattern_gbp_per_s": round(n * L / t / 1e9, 5),
                "reference_sample": "compiled reference (oracle/_ref, g++ -O3, 1 thread): BasePattern constructor on %d x %d bp in %.2f s" % (n, L, t)}
    except Exception as e:  # the probe is optional
        return {"reference_sample": "probe failed: %r" % (e,)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def reference_cli_phases(W, both, L, n=300_000):
    """The compiled reference CLI (oracle/_ref/peng_motif_ref) on a sample, phases from the arrival times of its stdout
    banners (src/peng.cpp:315-320): ingest + background model | BasePattern (bgprob, count, stats) | seed selection |
    IUPAC optimisation | filter + PWMs | EM + merging + output.  Once with --threads 1 and once with all cores (its
    OpenMP loops are the sweeps and the hill-climb; count and EM are serial)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "peng_motif_ref")
    if not os.path.exists(exe):
        return {"reference_cli_phases": "oracle/_ref/peng_motif_ref not present on this box"}
    import shutil
    import tempfile
    tmp = tempfile.mkdtemp(prefix="pengk_refcli_")
    # a banner is printed when the phase it names STARTS: the time up to banner i belongs to the phase of banner i - 1
    # ("Finding overrepresented kmers" precedes the BasePattern constructor and the seed selection)
    marks = [("[STATUS] Processing kmers", "ingest_x2_and_background_model"), ("[STATUS] Finding overrepresented", None),
             ("[STATUS] Optimizing base patterns", "basepattern_bgprob_count_stats_and_seed_selection"),
             ("[STATUS] Filtering degenerated", "iupac_optimisation"), ("[STATUS] Calculating PWMs", "filter"),
             ("[STATUS] Optimizing expectation", "pwm_construction"), ("merge:", "em")]
    try:
        fa = os.path.join(tmp, "s.fa")
        _write_sample_fasta(fa, n, L)
        res = {}
        cores = os.cpu_count() or 1
        for threads in (1, cores):
            t0 = time.perf_counter()
            p = subprocess.Popen([exe, fa, "-w", str(W), "--strand", "BOTH" if both else "PLUS", "--threads", str(threads), "-o",
                                  os.path.join(tmp, "o.meme")], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
            last, phases, idx = t0, {}, 0
            for raw in p.stdout:
                line = raw.decode(errors="replace")
                if idx < len(marks) and line.startswith(marks[idx][0]):
                    now = time.perf_counter()
                    if marks[idx][1]:
                        phases[marks[idx][1]] = round(now - last, 3)
                        last = now
                    idx += 1
            p.wait(timeout=900)
            now = time.perf_counter()
            phases["em_merging_output" if idx < len(marks) else "merging_output"] = round(now - last, 3)
            phases["total"] = round(now - t0, 3)
            res["threads_%d" % threads] = phases
            if threads == cores:
                break
        return {"reference_cli_phases": res, "reference_cli_sample": "compiled reference CLI on %d x %d bp of the same synthetic set, W=%d, host cores %d"
                % (n, L, W, cores)}
    except Exception as e:
        return {"reference_cli_phases": "failed: %r" % (e,)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
