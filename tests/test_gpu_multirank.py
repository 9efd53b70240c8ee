"""The multi-rank peng_motif, rehearsed on ONE GPU: two (and three) CLI processes started with a launcher's environment,
each reading only its byte range of the FASTA file (sharded ingest), counting its shard, summing {counts, ltot} and
gathering the EM's PWMs -- with PENGK_COMM_TRANSPORT=tcp, the rehearsal transport that stages the tables through host
memory over the host channel, because RCCL cannot put two ranks on one card.  Everything except the carrier of the sum
is the code an 8-GPU run executes (BASELINE configs[3]: "seqs sharded 8x MI355X, all-reduce of the 4^W counts").

Bar: stdout, MEME and JSON of rank 0 byte-identical to the plain single-process run; the other ranks print nothing on
stdout; stderr warnings identical; per-rank resident memory ~ 1/world of the file.
Reference for the single-process semantics: src/shared/SequenceSet.cpp:285-447, src/shared/BackgroundModel.cpp:60-84,
src/base_pattern.cpp:382 (the non-overlap rule never crosses a sequence boundary: shard counts add exactly)."""
import os
import re
import socket
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "peng-motif_amd", "host", "peng_motif")
GOLD = os.path.join(ROOT, "tests", "golden")
SYNTH = os.path.join(ROOT, "tools", "synth_fasta")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PENGK_COMM_TRANSPORT")}
    env.update(extra)
    return env


def run_plain(args, tmp_path, tag="plain", timing=False, extra_env=None):
    meme, js = tmp_path / (tag + ".meme"), tmp_path / (tag + ".json")
    env = clean_env(**({"PENGK_TIMING": "1"} if timing else {}), **(extra_env or {}))
    r = subprocess.run([CLI] + args + ["-o", str(meme), "-j", str(js)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env,
                       timeout=900)
    return r.returncode, r.stdout, r.stderr, meme.read_bytes() if meme.exists() else None, js.read_bytes() if js.exists() else None


def run_ranks(args, world, tmp_path, tag="ranks", timing=False, transport="tcp", extra_env=None):
    port = free_port()
    procs = []
    for rank in range(world):
        meme, js = tmp_path / ("%s%d.meme" % (tag, rank)), tmp_path / ("%s%d.json" % (tag, rank))
        env = clean_env(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                        PENGK_COMM_TRANSPORT=transport, PENGK_COMM_TIMEOUT="300")
        if timing:
            env["PENGK_TIMING"] = "1"
        env.update(extra_env or {})
        procs.append((subprocess.Popen([CLI] + args + ["-o", str(meme), "-j", str(js)], stdout=subprocess.PIPE,
                                       stderr=subprocess.PIPE, env=env), meme, js))
    out = []
    for p, meme, js in procs:
        so, se = p.communicate(timeout=900)
        out.append((p.returncode, so, se, meme.read_bytes() if meme.exists() else None, js.read_bytes() if js.exists() else None))
    return out


def strip_timing(stderr):
    return b"".join(l for l in stderr.splitlines(True) if not l.startswith(b"[timing]"))


def peak_rss_mb(stderr):
    m = re.search(rb"\[timing\] peak resident memory: ([0-9.]+) MiB", stderr)
    assert m, stderr[-500:]
    return float(m.group(1))


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("args", [["MafK.fasta", "-w", "10"], ["MafK.fasta", "-w", "8", "--strand", "PLUS"],
                                  ["torture.fa", "-w", "6"]], ids=["mafk_w10", "mafk_w8_plus", "torture_w6"])
def test_ranks_on_one_gpu_print_what_one_process_prints(tmp_path, args, world):
    args = [os.path.join(GOLD, args[0])] + args[1:]
    rc, so, se, meme, js = run_plain(args, tmp_path)
    assert rc == 0, se.decode()[-2000:]
    res = run_ranks(args, world, tmp_path)
    for rank, (rrc, rso, rse, rmeme, rjs) in enumerate(res):
        assert rrc == 0, (rank, rse.decode()[-2000:])
        if rank == 0:
            assert rso == so and rmeme == meme and rjs == js
            assert rse == se  # the reader's warnings, once, in file order
        else:
            assert rso == b"" and rmeme is None and rjs is None and rse == b""


def test_ranks_with_a_separate_background_file(tmp_path):
    args = [os.path.join(GOLD, "MafK_100seqs.fasta"), "-w", "8", "--background-sequences", os.path.join(GOLD, "MafK.fasta")]
    rc, so, se, meme, js = run_plain(args, tmp_path)
    assert rc == 0
    res = run_ranks(args, 2, tmp_path)
    assert [r[0] for r in res] == [0, 0], res[0][2].decode()[-1000:]
    assert (res[0][1], res[0][3], res[0][4]) == (so, meme, js) and res[1][1] == b""


def test_two_ranks_on_a_synthetic_set_and_their_memory(tmp_path):
    """300 000 x 200 bp (63 MB of FASTA), W=10: outputs byte-identical, and a rank's peak resident memory grows with ITS
    shard, not with the file -- measured as the difference to the same run on a tenth of the records."""
    fa = str(tmp_path / "s300k.fa")
    subprocess.check_call([SYNTH, fa, "300000", "200", "1", "0"])
    small = str(tmp_path / "s30k.fa")
    subprocess.check_call([SYNTH, small, "30000", "200", "1", "0"])
    rc, so, se, meme, js = run_plain([fa, "-w", "10"], tmp_path, timing=True)
    assert rc == 0, se.decode()[-2000:]
    res = run_ranks([fa, "-w", "10"], 2, tmp_path, timing=True)
    assert [r[0] for r in res] == [0, 0], res[0][2].decode()[-2000:]
    assert (res[0][1], res[0][3], res[0][4]) == (so, meme, js)
    assert strip_timing(res[0][2]) == strip_timing(se) and res[1][1] == b""
    # memory: with the staged path (PENGK_NO_STREAMING=1), which holds a shard's byte codes to the end -- the streaming
    # path gives every chunk's codes back as soon as they are packed, and its peak says little about the shard
    # Every figure is the SMALLER peak of three runs: a process's peak carries transients that have nothing to do with its
    # shard (allocator arenas of the reader's threads, the HIP runtime's staging buffers: +100 MB one run in three on a
    # busy box), and they only ever add.
    keep = {"PENGK_NO_STREAMING": "1"}
    REPEATS = 3
    peaks = {"whole": [], "whole_small": [], "ranks": [[], []], "base": [[], []]}
    for rep in range(REPEATS):
        whole = run_plain([fa, "-w", "10"], tmp_path, tag="plainkeep%d" % rep, timing=True, extra_env=keep)
        ranks = run_ranks([fa, "-w", "10"], 2, tmp_path, tag="rankskeep%d" % rep, timing=True, extra_env=keep)
        assert (ranks[0][1], ranks[0][3], ranks[0][4]) == (so, meme, js)
        base = run_ranks([small, "-w", "10"], 2, tmp_path, tag="small%d" % rep, timing=True, extra_env=keep)
        whole_small = run_plain([small, "-w", "10"], tmp_path, tag="plainsmall%d" % rep, timing=True, extra_env=keep)
        peaks["whole"].append(peak_rss_mb(whole[2]))
        peaks["whole_small"].append(peak_rss_mb(whole_small[2]))
        for rank in range(2):
            peaks["ranks"][rank].append(peak_rss_mb(ranks[rank][2]))
            peaks["base"][rank].append(peak_rss_mb(base[rank][2]))
    grow_plain = min(peaks["whole"]) - min(peaks["whole_small"])
    for rank in range(2):
        grow = min(peaks["ranks"][rank]) - min(peaks["base"][rank])
        # one process: file text + codes + packed stream for 270k more records; a rank: for 135k more
        assert grow < 0.65 * grow_plain, (rank, grow, grow_plain, peaks)


def test_a_fasta_error_in_one_shard_ends_every_rank(tmp_path):
    fa = tmp_path / "bad.fa"
    rng = np.random.default_rng(5)
    recs = [">r%d\n%s\n" % (i, "".join(rng.choice(list("ACGT"), 80).tolist())) for i in range(400)]
    recs[333] = ">r333\nACGT ACGT\n"
    fa.write_text("".join(recs))
    rc, so, se, _, _ = run_plain([str(fa), "-w", "6"], tmp_path)
    assert rc == 1 and b"contains space character" in se
    res = run_ranks([str(fa), "-w", "6"], 2, tmp_path)
    assert [r[0] for r in res] == [1, 1]
    assert res[0][2] == se and res[1][2] == b""


def test_a_rank_that_never_arrives_is_an_error_not_a_hang(tmp_path):
    port = free_port()
    env = clean_env(RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                    PENGK_COMM_TRANSPORT="tcp", PENGK_COMM_TIMEOUT="3")
    r = subprocess.run([CLI, os.path.join(GOLD, "MafK_100seqs.fasta"), "-w", "8"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=env, timeout=120)
    assert r.returncode == 1 and b"only 1 of 2 ranks" in r.stderr


@pytest.mark.parametrize("args", [["torture.fa", "-w", "6"], ["MafK.fasta", "-w", "10"], ["MafK.fasta", "-w", "12", "--strand", "PLUS"]],
                         ids=["torture_w6", "mafk_w10", "mafk_w12_plus"])
def test_streaming_pack_equals_the_staged_path(tmp_path, args):
    """The CLI packs and uploads every chunk of the input while the file is still being read (host/device.cpp) and takes
    the background counters from the packer; PENGK_NO_STREAMING=1 keeps the staged order read -> background model ->
    pack -> upload.  Same bytes on stdout / stderr / MEME / JSON, also with the reader forced into many small chunks on
    several threads (chunks then arrive out of order; torture.fa sends its chunks down the packer's general path)."""
    args = [os.path.join(GOLD, args[0])] + args[1:]
    outs = []
    for tag, extra in (("staged", {"PENGK_NO_STREAMING": "1"}), ("streamed", {}),
                       ("chunks", {"PENGK_READ_CHUNKS": "23", "PENGK_READ_THREADS": "5"}),
                       ("chunks_staged", {"PENGK_READ_CHUNKS": "7", "PENGK_READ_THREADS": "3", "PENGK_NO_STREAMING": "1"})):
        meme, js = tmp_path / (tag + ".meme"), tmp_path / (tag + ".json")
        r = subprocess.run([CLI] + args + ["-o", str(meme), "-j", str(js)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=clean_env(**extra), timeout=900)
        assert r.returncode == 0, (tag, r.stderr.decode()[-2000:])
        outs.append((r.stdout, r.stderr, meme.read_bytes(), js.read_bytes()))
    assert outs[0] == outs[1] == outs[2] == outs[3]


@pytest.mark.parametrize("text", [b"", b">a\n>b\n", b">a\nACGT\n>b\nAC\n", b">a\nNNNNNNNNNNNNNNNNNNNNNNNNNNNNNN\n>b\nNNNNNNNNNNNNNNNNN\n",
                                  b">a\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n", b">a\nACGTTGCAAGCTAGCTAGGATCGATCGATTAGCTAGCTAGCTAGGGATCGA"],
                         ids=["empty", "headers_only", "short", "all_n", "one_record", "no_newline"])
def test_ranks_on_degenerate_inputs(tmp_path, text):
    """Shards without a single record, without a single window, a file smaller than the number of ranks: three ranks
    still end like one process does (same exit code, stdout, stderr, MEME)."""
    fa = tmp_path / "d.fa"
    fa.write_bytes(text)
    rc, so, se, meme, js = run_plain([str(fa), "-w", "8"], tmp_path)
    res = run_ranks([str(fa), "-w", "8"], 3, tmp_path)
    assert [r[0] for r in res] == [rc] * 3, [r[2].decode()[-300:] for r in res]
    assert (res[0][1], res[0][2], res[0][3], res[0][4]) == (so, se, meme, js)
    assert res[1][1] == res[2][1] == b""


def test_a_rank_without_rccl_ends_every_rank_before_anyone_enters_comminit(tmp_path):
    """RCCL transport, two ranks, rank 1 cannot load librccl (PENGK_COMM_TEST_FAIL_LOAD): every rank reports its
    load_rccl() result over the host channel BEFORE anybody enters ncclCommInitRank (which has no deadline of its own),
    so both ranks end with the error at once -- rank 0 naming rank 1 -- instead of rank 0 waiting inside RCCL."""
    port = free_port()
    import time
    t0 = time.time()
    procs = []
    for rank in range(2):
        env = clean_env(RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                        PENGK_COMM_TRANSPORT="rccl", PENGK_COMM_TIMEOUT="60", PENGK_COMM_TEST_FAIL_LOAD="1")
        procs.append(subprocess.Popen([CLI, os.path.join(GOLD, "MafK_100seqs.fasta"), "-w", "8"], stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, env=env))
    res = [p.communicate(timeout=120) + (p.returncode,) for p in procs]
    assert time.time() - t0 < 50  # far inside the 60 s deadline: nobody waited for anybody
    assert [r[2] for r in res] == [1, 1], res
    assert b"rank 1 could not load librccl" in res[0][1], res[0][1][-600:]
    assert b"forced by PENGK_COMM_TEST_FAIL_LOAD" in res[1][1], res[1][1][-600:]


def test_a_rank_that_dies_in_front_of_comminit_ends_the_other_within_the_deadline(tmp_path):
    """RCCL transport, two ranks: rank 1 dies AFTER every rank has agreed that it can create the communicator and before it
    enters ncclCommInitRank (PENGK_COMM_TEST_DIE_BEFORE_INIT).  Rank 0 is then inside ncclCommInitRank -- which has no
    deadline of its own -- with a peer that never comes: its helper thread stays there, the call returns after
    PENGK_COMM_TIMEOUT seconds, and peng_motif leaves with _exit(1) (no exit handlers under the live thread inside RCCL:
    host/device.cpp, pengk_comm_init_abandoned).  Both ranks are gone, non-zero, well inside the test's limit."""
    port = free_port()
    import time
    t0 = time.time()
    procs = []
    for rank in range(2):
        env = clean_env(RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                        PENGK_COMM_TRANSPORT="rccl", PENGK_COMM_TIMEOUT="6", PENGK_COMM_TEST_DIE_BEFORE_INIT="1")
        procs.append(subprocess.Popen([CLI, os.path.join(GOLD, "MafK_100seqs.fasta"), "-w", "8"], stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, env=env))
    res = [p.communicate(timeout=120) + (p.returncode,) for p in procs]
    took = time.time() - t0
    assert res[1][2] == 3, res[1]
    assert res[0][2] == 1, res[0]
    assert took < 60, took  # the deadline (6 s) plus the start of two processes, not a hang
    assert b"pengk_comm_init_env failed" in res[0][1], res[0][1][-600:]


def test_comminit_has_a_deadline():
    """pengk_comm_init with an id whose other rank never joins: an error after PENGK_COMM_TIMEOUT seconds, not a process
    parked inside ncclCommInitRank (the call runs on a helper thread; the process leaves with os._exit, as a rank does
    after any communicator error)."""
    code = r"""
import ctypes as C, os, sys, time
sys.path.insert(0, %r)
import peng_motif_amd as pk
L = pk.lib()
ctx = pk.Context(0)
buf = C.create_string_buffer(128)
pk._check(L.pengk_comm_unique_id(buf))
t0 = time.time()
before = L.pengk_comm_init_abandoned()
rc = L.pengk_comm_init(ctx.h, buf.raw, 0, 2)
print("RC", rc, "%%.1f" %% (time.time() - t0), L.pengk_last_error().decode(), flush=True)
print("ABANDONED", before, L.pengk_comm_init_abandoned(), "VERSION", L.pengk_comm_rccl_version(), flush=True)
os._exit(0)
""" % ROOT
    import sys
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=clean_env(PENGK_COMM_TIMEOUT="4"), timeout=180)
    out = r.stdout.decode()
    line = [l for l in out.splitlines() if l.startswith("RC ")]  # (librccl announces its version on stdout as well)
    assert r.returncode == 0 and len(line) == 1, (out, r.stderr.decode()[-800:])
    rc, secs = line[0].split()[1:3]
    assert int(rc) != 0 and 3.5 <= float(secs) < 30 and "still waiting for the other ranks" in line[0], out
    # the process knows that a helper thread of it is still inside ncclCommInitRank (peng_motif then leaves with _exit,
    # host/device.cpp: check), and which NCCL API the bound librccl implements (>= 2.0, or it would have been refused)
    ab, = [l.split() for l in out.splitlines() if l.startswith("ABANDONED ")]
    assert ab[1] == "0" and ab[2] == "1" and int(ab[4]) >= 2000, ab


def _prefix_row(job, k):
    import json
    rows = json.load(open(os.path.join(GOLD, "shard_prefix_checksums.json")))
    row, = [r for r in rows if r["job"] == job and r["kind"] == "prefix" and r["k"] == k]
    return row


@pytest.mark.parametrize("job,k,world", [("configs2_weak", 2, 2), ("configs2_weak", 4, 4), ("configs3", 2, 2), ("configs3", 4, 4)])
def test_ranks_produce_the_sum_of_the_references_shard_tables(tmp_path, job, k, world):
    """The multi-rank CLI against the COMPILED REFERENCE, not against this program's own single-process run: the first k
    shards of BASELINE configs[3] (k x 12.5M x 200 bp, W = 12) / of bench.py's weak-scaled configs[2] sets (k x 10M x 200 bp,
    W = 10) as one FASTA file, read by `world` ranks (byte ranges), counted, summed over the exchange step; rank 0's global
    count table, ltot, V and z must be what the reference's per-shard tables add up to
    (tests/golden/shard_prefix_checksums.json; V and z: the oracle's sweep on those sums)."""
    import hashlib
    import shutil
    row = _prefix_row(job, k)
    need_gb = row["n_seq"] * (row["L"] + 12) / 1e9 * 1.1
    if shutil.disk_usage(str(tmp_path)).free / 1e9 < need_gb + 2:
        pytest.skip("needs %.0f GB of scratch for the FASTA file" % need_gb)
    fa = str(tmp_path / "set.fa")
    subprocess.check_call([SYNTH, fa, str(row["n_seq"]), str(row["L"]), str(row["seed"]), "0"])
    dump = tmp_path / "tables"
    dump.mkdir()
    args = [fa, "-w", str(row["W"])] + (["--strand", "PLUS"] if row["strand"] == "PLUS" else [])
    res = run_ranks(args, world, tmp_path, extra_env={"PENGK_DUMP_TABLES": str(dump)})
    os.remove(fa)
    assert [r[0] for r in res] == [0] * world, res[0][2].decode()[-2000:]
    sha = lambda name: hashlib.sha256((dump / name).read_bytes()).hexdigest()  # noqa: E731
    meta = dict(l.split() for l in (dump / "meta.txt").read_text().splitlines())
    assert int(meta["ltot"]) == row["ltot"] and int(meta["N"]) == row["n_seq"]
    assert sha("counts.u32") == row["sha_counts_u32"]
    assert sha("V.f32") == row["sha_V"]
    assert sha("expected.f32") == row["sha_expected"]
    assert sha("z.f32") == row["sha_z"]
