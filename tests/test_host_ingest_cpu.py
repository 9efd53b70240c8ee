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


def test_seed_ranking_replays_std_sort():
    """select_base_patterns ranks with ranked_prefix.h (a replay of libstdc++'s introsort that stops below the z
    threshold); the reference ranks with std::sort (src/base_pattern.cpp:458).  The CPU-only harness compares the two on
    this machine's libstdc++ for tables full of exact ties (reverse-complement pairs), +inf and NaN scores."""
    host = os.path.dirname(DUMP)
    subprocess.run(["make", "-s", "-C", host, "ranked_prefix_test"], check=True)
    r = subprocess.run([os.path.join(host, "ranked_prefix_test")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stdout.decode()
    assert r.stdout.decode().startswith("ok ")
