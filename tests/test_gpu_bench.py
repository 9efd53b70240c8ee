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
                        "--cpu-sample-seqs", "20000", "--em-stress-pwms", "32"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600)
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
    assert d["value"] > 0 and abs(d["value"] - 200000 * 200 / (d["ms_per_step"] * 1e-3) / 1e9) < 1e-2 * d["value"]


def test_two_rank_rehearsal_allreduces_the_tables():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, PENGK_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--nseq", "150000", "--em-stress-pwms", "8"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=env, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "cpu_baseline" not in d
    assert d["config"]["ltot_global"] == 2 * 150000 * 191  # both shards arrived in the reduced ltot
    assert abs(d["value"] - 2 * 150000 * 200 / (d["ms_per_step"] * 1e-3) / 1e9) < 1e-2 * d["value"]
