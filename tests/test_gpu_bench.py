"""bench.py contract on the GPU box: the single-rank JSON line, and a 2-rank rehearsal of the N > 1 path
(gloo transport, both ranks on the one GPU of the box; the measured multi-GPU runs use nccl = RCCL)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_json(out):
    lines = [l for l in out.decode().split("\n") if l.startswith("{")]
    assert len(lines) == 1, out.decode()[-2000:]
    return json.loads(lines[0])


def test_single_rank_json_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--nseq", "200000",
                        "--cpu-sample-seqs", "20000", "--em-stress-pwms", "32", "--config3-nseq", "50000", "--config3-steps", "2",
                        "--e2e-runs", "3"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = last_json(r.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["config"]["ltot_global"] == 200000 * 191 and "workload" in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-6
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0
    assert cb["em_stress_check"]["ok"], cb["em_stress_check"]  # configs[4] seeds: PWM 0 against the fp64 oracle
    for phase in ("count", "bgprob_tables_3_orders", "stats_sweep", "em_per_pwm_10_iterations"):
        assert cb["port_phases_s"][phase] > 0
    ref = cb["reference_cli_phases"]["threads_1"]  # the compiled reference travels with the repository (oracle/_ref)
    assert ref["total"] > 0 and "ingest_x2_and_background_model" in ref
    assert d["value"] > 0 and abs(d["value"] - 200000 * 200 / (d["ms_per_step"] * 1e-3) / 1e9) < 1e-2 * d["value"]
    assert "serial" in d["config"]["em_mode"] and d["components"]["em_other_mode"] == 1  # the step runs the CLI's EM mode
    assert d["roofline_em"]["flop_per_eval"] == 24 and d["roofline_em"]["frac"] > 0
    e2e = d["components"]["e2e_cli"]
    assert "error" not in e2e and e2e["wall_s"] > 0 and e2e["motifs"] >= 1 and "total" in e2e["phases_s"]
    # the spread is on the line: every run's wall time, the median as wall_s, where the runtime's start and the exit go
    assert len(e2e["walls_s"]) == 3 and min(e2e["walls_s"]) <= e2e["wall_s"] <= max(e2e["walls_s"]) and e2e["best_wall_s"] == min(e2e["walls_s"])
    assert e2e["exit_s"] is not None and e2e["exit_s"] >= 0 and any("runtime start" in k for k in e2e["runtime_start_ms"])
    assert 2 <= len(e2e["device_ready_probe_first_stream_ms"]) <= 12  # (untimed probe runs until the box answers normally)
    # the exchange step has its own events; the checks are always on the line (no reference-derived row at this toy size)
    assert d["components"]["exchange_ms"] >= 0 and set(d["checks"]) == {"sha_counts", "sha_z", "sha_bg_ltot", "sha_em_pwms"}
    assert d["checks_ok"]["ok"] is None
    c3 = d["components"]["config3"]
    assert c3["ltot_global"] == 50000 * 189 and c3["count_ms"] > 0 and c3["exchange_bytes"] == 4 * 4 ** 12 + 680 and c3["checks_ok"]["ok"] is None
    pl = d["components"]["pipelined"]  # the same passes as a two-stream pipeline: an extra throughput figure, same bits
    assert pl["same_bits_as_the_sequential_passes"] is True and pl["ms_per_step"] > 0 and pl["steps"] == 2
    c4 = d["components"]["config4"]
    assert c4["pwms"] == 32 and c4["split_equals_1rank_bit_for_bit"] is True and c4["ms_split"]["serial"] > 0 and c4["allgather_ms"] >= 0
    # the compiled reference on the same file, same box, same run: the denominator of the end-to-end ratio -- and its
    # MEME output is this repository's byte for byte
    same = e2e["reference_same_box"]
    if "skipped" in same:
        pytest.skip("reference_same_box: " + same["skipped"])
    assert same["reference_wall_s_same_box"] > 0 and same["speedup_same_box"] > 0 and same["meme_identical_to_reference"] is True
    assert cb["reference_full_size"] == same
    # labels: only BASELINE.json's own sizes carry its config names; traffic only from a profile of this configuration
    assert "not a BASELINE.json configuration" in d["config"]["workload"]
    assert rf["traffic"] is None and "roofline_issue" not in d
    assert d["roofline_em"]["parity_mode"]["frac"] > 0 and d["roofline_em"]["parity_mode"]["ms"] > 0
    # round 5: how the exchange ran is on the top level (one rank: no communicator, no fallback); the sweep and K4 carry
    # their own roofline objects (28 B per pattern, 16 B per visited k-mer: SURVEY.md 8d); the strong-scaled leg at one rank
    # is the line's own step; the end-to-end leg says where its FASTA lay and what the first run cost
    assert d["rccl_ranks"] == 0 and d["exchange_fallback"] is None
    for key, alg in (("roofline_sweep", 28 * 4 ** 10), ):
        r_ = d[key]
        assert r_["bound"] == "hbm" and r_["algorithmic_bytes_per_launch"] == alg and abs(r_["frac"] - r_["achieved"] / r_["peak"]) < 1e-4
    assert d["roofline_k4"]["bound"] == "hbm" and d["roofline_k4"]["achieved"] > 0 and d["roofline_k4"]["algorithmic_bytes_per_call"] % 16 == 0
    assert c3["roofline_sweep"]["algorithmic_bytes_per_launch"] == 28 * 4 ** 12 and "weak" in c3["scaling"] and "strong" in c4["scaling"]
    st = d["components"]["config2_strong"]
    assert st["scaling"] == "strong" and st["n_gpus"] == 1 and st["speedup_vs_1rank"] == 1.0 and st["ms_per_step"] == d["ms_per_step"]
    assert e2e["fasta_on"] in ("tmpfs", "disk") and e2e["cold_first_run_s"] == e2e["walls_s"][0] and same["speedup_cold"] > 0
    al = cb["sweeps_all_threads"]
    assert al["threads"] >= 1 and al["zscores_per_s"] > 0 and al["bgprob_tables_3_orders_s"] > 0


def test_two_rank_rehearsal_allreduces_the_tables():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, PENGK_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--nseq", "150000", "--em-stress-pwms", "8", "--config3-steps", "0"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=env, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "cpu_baseline" not in d
    assert d["config"]["ltot_global"] == 2 * 150000 * 191  # both shards arrived in the reduced ltot
    assert abs(d["value"] - 2 * 150000 * 200 / (d["ms_per_step"] * 1e-3) / 1e9) < 1e-2 * d["value"]
    assert d["components"]["pipelined"]["same_bits_as_the_sequential_passes"] is True  # (also with the exchange step inside)
    c4 = d["components"]["config4"]  # 8 stress PWMs dealt to two ranks, gathered, equal to all 8 on one rank
    assert c4["n_gpus"] == 2 and c4["split_equals_1rank_bit_for_bit"] is True


def _bench(extra, env=None, launcher=None):
    cmd = ([sys.executable] + (launcher or []) + [os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--em-stress-pwms", "0",
                                                  "--k4-patterns", "0", "--no-cpu-baseline", "--no-e2e", "--config3-steps", "0"] + extra)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    return last_json(r.stdout)


def test_two_ranks_through_the_hip_path_reproduce_one_rank_bit_for_bit():
    """Two ranks, each counting its 150k-sequence shard ON THE GPU, after the exchange step hold the tables a single
    rank computes from the 300k sequences: sha256 of the count table, of {bg counters, ltot}, of z and of the gathered
    serial-mode EM PWMs are equal.  (gloo carries the exchange here -- RCCL cannot put two ranks on the box's one GPU;
    the RCCL calls themselves are covered by test_rccl_single_rank_communicator and by the N > 1 bench runs.)"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    two = _bench(["--gpus", "2", "--nseq", "150000"], env=dict(os.environ, PENGK_BENCH_BACKEND="gloo"),
                 launcher=["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                           "--master-port", str(port)])
    one = _bench(["--nseq", "300000"])
    assert two["config"]["ltot_global"] == one["config"]["ltot_global"] == 300000 * 191
    assert two["checks"] == one["checks"], (two["checks"], one["checks"])


def test_one_rank_through_the_rccl_code_path_of_the_bench():
    """The N > 1 path of bench.py -- nccl process group, communicator id over the store, pengk_comm_init, the bin-bound
    collective, pengk_allreduce_tables inside every step, the gathers -- with the one rank a one-GPU box can hold
    (PENGK_BENCH_FORCE_COMM=1): same tables as the plain run."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    forced = _bench(["--gpus", "1", "--nseq", "200000"], env=dict(os.environ, PENGK_BENCH_FORCE_COMM="1"),
                    launcher=["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                              "--master-port", str(port)])
    plain = _bench(["--nseq", "200000"])
    assert forced["config"]["exchange"].startswith("RCCL, 1 rank")
    assert forced["checks"] == plain["checks"]


def test_strong_scaled_leg_at_one_rank_through_the_communicator():
    """components.config2_strong with the one rank a one-GPU box can hold, through the N > 1 code path
    (PENGK_BENCH_FORCE_COMM=1): the set "split" one way, the exchange through the library's RCCL communicator, and beside
    it the same set on one rank without any exchange -- same bits, a ratio near 1, and the top-level flags say that the
    exchange really ran in the library."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    d = _bench(["--gpus", "1", "--nseq", "200000", "--strong", "1"], env=dict(os.environ, PENGK_BENCH_FORCE_COMM="1"),
               launcher=["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                         "--master-port", str(port)])
    assert d["rccl_ranks"] == 1 and d["exchange_fallback"] is None
    st = d["components"]["config2_strong"]
    assert st["scaling"] == "strong" and st["n_gpus"] == 1 and st["same_bits_as_1rank"]["ok"] is True, st
    assert st["ms_per_step"] > 0 and st["ms_1rank"] > 0 and 0.3 < st["speedup_vs_1rank"] < 3.0
    assert st["exchange_bytes"] == 4 * 4 ** 10 + 680


def test_strong_scaled_leg_on_two_ranks():
    """The 300k-sequence set split over two ranks (gloo rehearsal: both on the box's one GPU, the exchange through the
    host): after the exchange both hold the table one rank computes from the whole set -- counts, background counters, z
    and the PWMs of the EM, sha256 for sha256 --, and the line reports the two step times and their ratio."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    d = _bench(["--gpus", "2", "--nseq", "300000", "--strong", "1"], env=dict(os.environ, PENGK_BENCH_BACKEND="gloo"),
               launcher=["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", str(port)])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 0 and d["exchange_fallback"] is None and d["scaling"] == "weak"
    st = d["components"]["config2_strong"]
    assert st["scaling"] == "strong" and st["n_gpus"] == 2 and st["same_bits_as_1rank"]["ok"] is True, st
    assert all(st["same_bits_as_1rank"]["parts"].values()) and set(st["same_bits_as_1rank"]["parts"]) == {"sha_counts", "sha_z", "sha_bg_ltot", "sha_em_pwms"}
    assert st["ms_per_step"] > 0 and st["ms_1rank"] > 0 and st["speedup_vs_1rank"] > 0
    # the whole set's table is also what ONE rank of the weak layout holds for 300k sequences
    one = _bench(["--nseq", "300000"])
    assert one["checks"]["sha_counts"] and one["components"]["config2_strong"]["n_gpus"] == 1


def test_bench_exchange_safety_net_when_the_library_communicator_fails():
    """If pengk_comm_init fails on some rank at N > 1 the ranks agree (an all_reduce of a flag) to run the exchange as
    torch.distributed's RCCL all_reduce of the same device buffers, and the JSON line says so: forced here with the one
    rank of the test box (PENGK_BENCH_BREAK_COMM=1), same tables as the plain run."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    broken = _bench(["--gpus", "1", "--nseq", "200000"], env=dict(os.environ, PENGK_BENCH_FORCE_COMM="1", PENGK_BENCH_BREAK_COMM="1"),
                    launcher=["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                              "--master-port", str(port)])
    plain = _bench(["--nseq", "200000"])
    assert broken["config"]["exchange"].startswith("RCCL via torch.distributed all_reduce (fallback")
    assert broken["checks"] == plain["checks"]
    # ... and nobody can take such a line for one that went through the library's exchange
    assert broken["rccl_ranks"] == 0 and "PENGK_BENCH_BREAK_COMM" in broken["exchange_fallback"] and plain["exchange_fallback"] is None


def test_bench_starts_its_own_launcher_for_gpus_n():
    """`python bench.py --gpus 2` without a launcher starts torch.distributed.run itself (as a child process, before
    anything touches the GPU) and relays the one JSON line (gloo rehearsal: two ranks on the box's one GPU)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["PENGK_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--nseq", "100000",
                        "--em-stress-pwms", "0", "--k4-patterns", "0", "--config3-steps", "0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["ltot_global"] == 2 * 100000 * 191


def test_gpus_flag_must_match_the_launch():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "1"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"), timeout=300)
    assert r.returncode != 0 and b"--gpus 3" in r.stderr


def test_rccl_single_rank_communicator():
    """The C++ exchange step against the real librccl on the GPU box: a 1-rank communicator (all RCCL allows on one GPU)
    runs the grouped all-reduce, the bin-bound collective and the all-gather through RCCL and leaves the data unchanged."""
    import ctypes as C

    import numpy as np

    sys.path.insert(0, ROOT)
    import peng_motif_amd as pk
    lib = pk.lib()
    ctx = pk.Context(0)
    try:
        W = 8
        ctx.synth(3, 0, 5000, 120, W)
        counts, ltot, bg = ctx.count_bg(True)
        before = (counts.to_host().copy(), ltot.to_host().copy(), bg.to_host().copy())
        idb = C.create_string_buffer(128)
        pk._check(lib.pengk_comm_unique_id(idb))
        pk._check(lib.pengk_comm_init(ctx.h, idb, 0, 1))
        r_, w_ = C.c_int(-1), C.c_int(-1)
        pk._check(lib.pengk_comm_info(ctx.h, C.byref(r_), C.byref(w_)))
        assert (r_.value, w_.value) == (0, 1)
        pk._check(lib.pengk_comm_check_bin_bound(ctx.h))
        pk._check(lib.pengk_allreduce_tables(ctx.h, W, pk._ptr(counts), pk._ptr(ltot), pk._ptr(bg)))
        ctx.synchronize()
        assert np.array_equal(counts.to_host(), before[0]) and np.array_equal(ltot.to_host(), before[1])
        assert np.array_equal(bg.to_host(), before[2])
        src = pk.DeviceArray.from_host(ctx, np.arange(1000, dtype=np.float32))
        dst = ctx.empty(1000, np.float32)
        pk._check(lib.pengk_allgather(ctx.h, pk._ptr(src), pk._ptr(dst), 4000))
        ctx.synchronize()
        assert np.array_equal(dst.to_host(), np.arange(1000, dtype=np.float32))
        assert lib.pengk_comm_init(ctx.h, idb, 0, 1) != 0  # a context carries one communicator
        pk._check(lib.pengk_comm_destroy(ctx.h))
    finally:
        ctx.close()


def test_cli_under_a_launcher_environment_equals_the_plain_run(tmp_path):
    """peng_motif started with RANK=0 WORLD_SIZE=1 (a launcher's environment) builds its RCCL communicator, shards
    (one shard), all-reduces and all-gathers through RCCL -- and prints what the plain run prints, byte for byte."""
    cli = os.path.join(ROOT, "peng-motif_amd", "host", "peng_motif")
    fa = os.path.join(ROOT, "tests", "golden", "MafK.fasta")
    outs = []
    for tag, extra in (("plain", {}), ("launched", {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                                                     "MASTER_PORT": "29533"})):
        meme, js = tmp_path / (tag + ".meme"), tmp_path / (tag + ".json")
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
        env.update(extra)
        r = subprocess.run([cli, fa, "-w", "10", "-o", str(meme), "-j", str(js)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=env, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        outs.append((r.stdout, meme.read_bytes(), js.read_bytes()))
    assert outs[0] == outs[1]


def _full_size_bench(gpus, env=None):
    port_s = socket.socket()
    port_s.bind(("127.0.0.1", 0))
    port = port_s.getsockname()[1]
    port_s.close()
    cmd = [sys.executable] + (["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
                               "--master-port", str(port)] if env else []) + \
        [os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1", "--em-stress-pwms", "64", "--k4-patterns", "0",
         "--no-cpu-baseline", "--no-e2e", "--config3-steps", "2"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=1200)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    return last_json(r.stdout)


@pytest.mark.parametrize("gpus,how", [(1, "plain"), (1, "rccl"), (2, "gloo")])
def test_the_bench_line_verifies_itself_against_the_reference_at_full_size(gpus, how):
    """BASELINE configs[2] (10M x 200 bp per rank, W = 10) and the configs[3] leg (12.5M x 200 bp per rank, W = 12) at their
    real sizes: `checks_ok` compares what the ranks hold after the exchange step with the compiled reference's per-shard
    tables added up for that number of ranks (tests/golden/shard_prefix_checksums.json).  One rank, and two ranks on the
    box's one GPU with gloo as the carrier of the sum (rank r = shard r, as under RCCL on a multi-GPU node); and one rank
    through the N > 1 code path with a real RCCL communicator (PENGK_BENCH_FORCE_COMM=1)."""
    env = {"plain": None, "rccl": dict(os.environ, PENGK_BENCH_FORCE_COMM="1"), "gloo": dict(os.environ, PENGK_BENCH_BACKEND="gloo")}[how]
    d = _full_size_bench(gpus, env=env)
    if how == "rccl":
        assert d["config"]["exchange"].startswith("RCCL, 1 rank")
    assert d["n_gpus"] == gpus and d["checks_ok"]["ok"] is True, d["checks_ok"]
    assert "configs2_weak, k=%d" % gpus in d["checks_ok"]["against"]
    c3 = d["components"]["config3"]
    assert c3["checks_ok"]["ok"] is True and "configs3, k=%d" % gpus in c3["checks_ok"]["against"], c3["checks_ok"]
    assert c3["ltot_global"] == gpus * 12_500_000 * 189 and c3["exchange_bytes"] == 67109544
    r3 = c3["roofline"]  # the leg's own K1 roofline, traffic from the committed W = 12 profile
    assert r3["bound"] == "hbm" and abs(r3["frac"] - r3["achieved"] / r3["peak"]) < 1e-6 and r3["traffic"] > r3["algorithmic_bytes_per_launch"]
    assert d["components"]["config4"]["split_equals_1rank_bit_for_bit"] is True
