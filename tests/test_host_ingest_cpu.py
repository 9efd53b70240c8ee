"""Host ingest of the C++ mirror (mmap + threaded FASTA reader, threaded background counts) against the
oracle's restatement of the reference reader / BackgroundModel.  CPU only."""
import os
import re
import subprocess

import numpy as np
import pytest

from oracle import oracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DUMP = os.path.join(ROOT, "peng-motif_amd", "host", "host_ingest_dump")


def fnv(codes):
    h = 1469598103934665603
    for chunk in np.array_split(codes, max(1, len(codes) // 4_000_000)):
        for c in chunk.tolist() if len(codes) < 200_000 else []:
            h = ((h ^ c) * 1099511628211) & (2 ** 64 - 1)
    return h


def run_dump(path):
    if not os.path.exists(DUMP):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(DUMP), "host_ingest_dump"])
    r = subprocess.run([DUMP, path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-500:]
    out = r.stdout.decode().split("\n")
    head = dict(zip(out[0].split()[0::2], out[0].split()[1::2]))
    counts = np.array([int(x) for x in out[1].split()[1:]], np.int64)
    V = np.array([int(x, 16) for x in out[2].split()[1:]], np.uint32)
    return head, counts, V, out


@pytest.mark.parametrize("fasta", ["torture.fa", "MafK_100seqs.fasta", "MafK.fasta", "default_sequence_set.fa"])
def test_reader_and_background_model_match_oracle(golden_dir, fasta):
    path = os.path.join(golden_dir, fasta)
    head, counts, V, out = run_dump(path)
    codes, offs = po.read_fasta(path)
    lens = np.diff(offs)
    assert int(head["N"]) == len(lens) and int(head["total"]) == int(offs[-1])
    assert int(head["minL"]) == int(lens.min()) and int(head["maxL"]) == int(lens.max())
    if len(codes) < 200_000:
        assert int(head["fnv"], 16) == fnv(codes)
    n = po.bg_counts(codes, offs, 2)
    assert np.array_equal(counts, n)
    assert np.array_equal(V, po.bg_V(n, 2).view(np.uint32))
    assert out[3] == "views ok"


def test_threaded_reader_on_a_large_ragged_file(tmp_path):
    """> 4 MB so that every stage runs multi-threaded; ragged records, wrapped lines, N, blank lines,
    a '>'-only header, a header without sequence and an unterminated last line."""
    rng = np.random.default_rng(4)
    parts = []
    for i in range(30000):
        L = int(rng.integers(1, 400))
        s = rng.choice(list("ACGTacgtN"), size=L, p=[.23, .23, .23, .23, .015, .015, .015, .015, .02])
        s = "".join(s.tolist())
        hdr = ">" if i % 977 == 0 else ">r%d some text" % i
        body = "\n".join(s[j:j + 70] for j in range(0, L, 70))
        parts.append(hdr + "\n" + body + ("\n\n" if i % 501 == 0 else "\n"))
        if i % 1500 == 7:
            parts.append(">empty%d\n" % i)
    parts.append(">last\nACGTACGTAC")  # unterminated: dropped
    p = tmp_path / "big.fa"
    p.write_text("".join(parts))
    assert p.stat().st_size > (1 << 22)
    head, counts, V, out = run_dump(str(p))
    codes, offs = po.read_fasta(str(p))
    assert int(head["N"]) == len(offs) - 1 == 30000 and int(head["total"]) == int(offs[-1])
    n = po.bg_counts(codes, offs, 2)
    assert np.array_equal(counts, n)
    assert np.array_equal(V, po.bg_V(n, 2).view(np.uint32))
    assert out[3] == "views ok" and out[4] == "header0 1"


def test_reader_error_exits(tmp_path):
    for text in (">a\nAC GT\n", "ACGT\n>a\nAC\n"):
        p = tmp_path / "bad.fa"
        p.write_text(text)
        subprocess.run(["make", "-s", "-C", os.path.dirname(DUMP), "host_ingest_dump"], check=True)
        r = subprocess.run([DUMP, str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 1


def test_reader_on_paths_that_are_not_regular_files(tmp_path):
    """The reference reads through std::ifstream + getline (src/shared/SequenceSet.cpp:285-300): a DIRECTORY opens and
    yields no line -- an empty set, not an error (checked against the compiled reference: exit code 0, the empty file's
    stdout) --, a pipe is read to its end.  The mirror's pread-based reader treats a directory as an empty file and spools
    a pipe into memory: same dump as the same bytes in a file; a missing path is still the reference's error."""
    subprocess.run(["make", "-s", "-C", os.path.dirname(DUMP), "host_ingest_dump"], check=True)
    text = ">a\nACGTACGTACNNACGT\n>b\nGGGTTTAAcc\n>c\n" + "ACGTTGCA" * 5000 + "\n"
    f = tmp_path / "x.fa"
    f.write_text(text)
    want = subprocess.run([DUMP, str(f)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert want.returncode == 0
    fifo = tmp_path / "pipe.fa"
    os.mkfifo(fifo)
    p = subprocess.Popen([DUMP, str(fifo)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    with open(fifo, "w") as w:
        w.write(text)
    out, err = p.communicate(timeout=60)
    assert p.returncode == 0 and out == want.stdout
    with open(f, "rb") as src:
        r = subprocess.run([DUMP, "/dev/stdin"], stdin=src, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 0 and r.stdout == want.stdout
    empty = tmp_path / "empty.fa"
    empty.write_text("")
    d = tmp_path / "dir.fa"
    d.mkdir()
    a = subprocess.run([DUMP, str(empty)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    b = subprocess.run([DUMP, str(d)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert a.returncode == b.returncode == 0 and a.stdout == b.stdout and a.stdout.startswith(b"N 0 ")
    r = subprocess.run([DUMP, str(tmp_path / "missing.fa")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 1 and b"Cannot open FASTA file" in r.stderr


def test_seed_ranking_replays_std_sort():
    """select_base_patterns ranks with ranked_prefix.h (a replay of libstdc++'s introsort that stops below the z
    threshold); the reference ranks with std::sort (src/base_pattern.cpp:458).  The CPU-only harness compares the two on
    this machine's libstdc++ for tables full of exact ties (reverse-complement pairs), +inf and NaN scores."""
    host = os.path.dirname(DUMP)
    subprocess.run(["make", "-s", "-C", host, "ranked_prefix_test"], check=True)
    r = subprocess.run([os.path.join(host, "ranked_prefix_test")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stdout.decode()
    assert r.stdout.decode().startswith("ok ")


# ---- sharded ingest: every rank reads only its byte range (multi-GPU CLI; replaces the whole-file pass of
#      src/shared/SequenceSet.cpp:285-447 per rank) ---------------------------------------------------------------------
def run_sharded(path, world, tmp_path):
    """WORLD processes of host_ingest_dump, combining their shards through files (the stand-in for the host channel)."""
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(DUMP), "host_ingest_dump"])
    d = tmp_path / ("gather_w%d" % world)
    d.mkdir()
    procs = [subprocess.Popen([DUMP, path, str(r), str(world), str(d), str(d / ("codes%d" % r))], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE) for r in range(world)]
    outs = [p.communicate(timeout=300) for p in procs]
    return [p.returncode for p in procs], outs, d


def parse_dump(stdout):
    out = stdout.decode().split("\n")
    t = out[0].split()
    return dict(zip(t[0::2], t[1::2])), out


def ragged_fasta(tmp_path, n=6000, seed=11, tail=">last\nACGNNTRACGTAC\n"):
    rng = np.random.default_rng(seed)
    parts = ["\n\n"]  # blank lines in front of the first header are allowed
    for i in range(n):
        L = int(rng.integers(1, 300)) if i % 97 else int(rng.integers(3000, 9000))  # some records span several cuts
        s = "".join(rng.choice(list("ACGTacgtN"), size=L, p=[.23, .23, .23, .23, .015, .015, .015, .015, .02]).tolist())
        hdr = ">" if i % 311 == 0 else ">r%d > not a header" % i
        parts.append(hdr + "\n" + "\n".join(s[j:j + 60] for j in range(0, L, 60)) + "\n")
        if i % 700 == 3:
            parts.append(">empty%d\n" % i)
    parts.append(tail)
    p = tmp_path / "ragged.fa"
    p.write_text("".join(parts))
    return str(p)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_ingest_equals_the_whole_file_reader(tmp_path, world):
    path = ragged_fasta(tmp_path)
    whole = subprocess.run([DUMP, path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert whole.returncode == 0
    head0, out0 = parse_dump(whole.stdout)
    rcs, outs, d = run_sharded(path, world, tmp_path)
    assert rcs == [0] * world, [o[1].decode()[-300:] for o in outs]
    codes, offs = po.read_fasta(path)
    got_codes, got_lens, base = [], [], 0
    for r, (so, se) in enumerate(outs):
        head, out = parse_dump(so)
        # global quantities are the whole file's on every rank: N, min / max length, base frequencies, counters, V
        for key in ("N", "minL", "maxL", "f0", "f1", "f2", "f3"):
            assert head[key] == head0[key], (r, key)
        assert out[1] == out0[1] and out[2] == out0[2]
        assert int(head["base"]) == base
        base += int(head["localN"])
        got_codes.append(np.fromfile(str(d / ("codes%d" % r)), np.uint8))
        got_lens.append(np.diff(np.fromfile(str(d / ("codes%d.offs" % r)), np.int64)))
        # rank 0 alone speaks for the file, with the single reader's words
        assert se == (whole.stderr if r == 0 else b"")
    assert base == int(head0["N"]) == len(offs) - 1
    assert np.array_equal(np.concatenate(got_codes), codes)
    assert np.array_equal(np.concatenate(got_lens), np.diff(offs))
    assert whole.stderr.count(b"undefined base") == 3 and whole.stderr.count(b"without sequence") == 9


def test_sharded_ingest_names_an_unnamed_last_record_by_its_global_index(tmp_path):
    path = ragged_fasta(tmp_path, n=500, tail=">\nACGTNACGT\n")
    whole = subprocess.run([DUMP, path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    rcs, outs, _ = run_sharded(path, 4, tmp_path)
    assert rcs == [0] * 4 and outs[0][1] == whole.stderr
    assert b"undefined base: N at sequence 501" in whole.stderr


def test_sharded_ingest_with_more_ranks_than_records(tmp_path):
    p = tmp_path / "tiny.fa"
    p.write_text(">a\nACGTACGTAC\n>b\nGGGTTTAAAC\nCC")  # unterminated last line: dropped by the reader
    whole = subprocess.run([DUMP, str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    head0, out0 = parse_dump(whole.stdout)
    rcs, outs, _ = run_sharded(str(p), 5, tmp_path)
    assert rcs == [0] * 5
    total = 0
    for so, _ in outs:
        head, out = parse_dump(so)
        assert head["N"] == head0["N"] == "2" and out[1:3] == out0[1:3]
        total += int(head["localN"])
    assert total == 2


@pytest.mark.parametrize("text", [">a\nACGT\n" * 50 + ">b\nAC GT\n" + ">c\nACGT\n" * 50, "ACGT\n" + ">a\nACGT\n" * 100])
def test_sharded_ingest_errors_end_every_rank(tmp_path, text):
    """A format error in one rank's shard is every rank's error, reported once by rank 0 (the reference: exit(1))."""
    p = tmp_path / "bad.fa"
    p.write_text(text)
    whole = subprocess.run([DUMP, str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    rcs, outs, _ = run_sharded(str(p), 3, tmp_path)
    assert whole.returncode == 1 and rcs == [1, 1, 1]
    assert outs[0][1] == whole.stderr and outs[1][1] == outs[2][1] == b""
