"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/pengk.h declares,
refuses to run without a GPU (no fallback), and its host packer reproduces the reference's scan
rule (checked against the oracle's window walk).  No device compute here."""
import os
import re

import numpy as np
import pytest

import peng_motif_amd as pk
from oracle import oracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "pengk.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pengk_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = pk.lib()
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), n
    assert sorted(pk.EXPORTS) == names
    assert L.pengk_version() == 100


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pk.PengkError) as e:
        pk.Context(0)
    assert e.value.code == pk.ERR_DEVICE


def unpack_windows(p, both):
    """Replay the packed stream + items on the CPU: canonical id of every window, per run."""
    W = p.W
    words = p.words
    def base(g):
        return (int(words[g >> 5]) >> (2 * (g & 31))) & 3
    runs = []
    cur = None
    for rec in p.items.tolist():
        ws = rec & ((1 << 40) - 1)
        nw = (rec >> 40) & 0xFFFF
        cont = (rec >> 56) & 1
        if not cont:
            cur = []
            runs.append(cur)
        for t in range(nw):
            x = 0
            for q in range(W):
                x |= base(ws + t + q) << (2 * q)
            cur.append(min(x, po.revcomp(x, W)) if both else x)
    return runs


def greedy_count(runs, W, NP):
    counts = np.zeros(NP, np.uint64)
    for run in runs:
        last = {}
        for e, c in enumerate(run):
            if c not in last or last[c] + W <= e:
                counts[c] += 1
                last[c] = e
    return counts


@pytest.mark.parametrize("W,both", [(4, True), (6, True), (6, False), (8, False)])
def test_packer_reproduces_scan_rule(golden_dir, W, both):
    codes, offs = po.read_fasta(os.path.join(golden_dir, "torture.fa"))
    p = pk.Packed(codes, offs, W, 64)
    want, ltot = po.count(codes, offs, W, False)  # PLUS counts are un-mirrored, fine for comparison
    assert p.n_windows == ltot
    runs = unpack_windows(p, False)
    assert sum(len(r) for r in runs) == ltot
    got = greedy_count(runs, W, 4 ** W)
    assert np.array_equal(got, want)
    if both:
        want_b, _ = po.count(codes, offs, W, True)
        got_b = greedy_count(unpack_windows(p, True), W, 4 ** W)
        canon = np.array([x <= po.revcomp(x, W) for x in range(4 ** W)])
        assert np.array_equal(got_b[canon], want_b[canon])
    assert np.array_equal(p.bg_counts, po.bg_counts(codes, offs, 2))
    assert p.all_whole == 0
    assert p.max_bin_bound >= int(want.max())


def test_packer_items_and_flags(golden_dir):
    codes, offs = po.read_fasta(os.path.join(golden_dir, "MafK_100seqs.fasta"))
    p = pk.Packed(codes, offs, 8, 64)
    assert p.all_whole == 1 and p.n_sequences == 25 and p.max_len == 205
    # 198 windows per sequence -> 4 items of <= 64 windows, the last three continue the run
    assert len(p.items) == 25 * 4
    cont = (p.items >> np.uint64(56)) & np.uint64(1)
    assert cont.reshape(25, 4).tolist() == [[0, 1, 1, 1]] * 25
    nw = (p.items >> np.uint64(40)) & np.uint64(0xFFFF)
    assert nw.reshape(25, 4).sum(axis=1).tolist() == [198] * 25
    assert p.n_bases == 25 * 205 and p.n_windows == 4950
    assert np.array_equal(p.bg_counts, po.bg_counts(codes, offs, 2))
    # front pad is zero, stream starts at base 64
    assert p.words[0] == 0 and p.words[1] == 0
    first = [(int(p.words[2]) >> (2 * i)) & 3 for i in range(8)]
    assert first == (codes[:8] - 1).tolist()


def test_packer_argument_errors():
    codes = np.array([1, 2, 3, 4] * 5, np.uint8)
    offs = np.array([0, 20], np.int64)
    for W in (3, 5, 16, 0):  # (W = 2 is a pattern length since round 4: the reference accepts every even W)
        with pytest.raises(pk.PengkError) as e:
            pk.Packed(codes, offs, W)
        assert e.value.code == pk.ERR_ARG
    with pytest.raises(pk.PengkError):
        pk.Packed(codes, offs, 8, 8)  # item_windows below the minimum
    p2 = pk.Packed(codes, offs, 2)
    assert p2.n_windows == 19 and p2.W == 2
    p = pk.Packed(codes[:0], np.array([0], np.int64), 8)  # empty input is fine
    assert p.n_windows == 0 and len(p.items) == 0


def test_synth_sizes():
    import ctypes as C
    nw, ni = C.c_uint64(), C.c_uint64()
    assert pk.lib().pengk_synth_sizes(1000, 200, 10, 0, C.byref(nw), C.byref(ni)) == 0
    assert ni.value == 1000 and nw.value == (64 + 200000 + 31) // 32 + 4
    assert pk.lib().pengk_synth_sizes(10, 300, 10, 64, C.byref(nw), C.byref(ni)) == 0
    assert ni.value == 10 * 5
    assert pk.lib().pengk_synth_sizes(10, 8, 10, 64, C.byref(nw), C.byref(ni)) == pk.ERR_ARG


def test_packer_is_thread_count_invariant(monkeypatch):
    """the two-pass threaded packer writes the same stream, items and counters for any number of host threads"""
    rng = np.random.default_rng(9)
    lens = rng.integers(1, 400, size=3000)
    codes = rng.integers(1, 5, size=int(lens.sum())).astype(np.uint8)
    codes[rng.integers(0, len(codes), size=400)] = 0  # invalid bases split runs
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ref = None
    for nt in (1, 2, 7, 16):
        monkeypatch.setenv("PENGK_PACK_THREADS", str(nt))
        p = pk.Packed(codes, offs, 8, 64)
        cur = (p.words.tobytes(), p.items.tobytes(), p.bg_counts.tobytes(), p.n_windows, p.max_bin_bound, p.all_whole, p.max_len)
        if ref is None:
            ref = cur
            want, ltot = po.count(codes, offs, 8, False)
            assert p.n_windows == ltot
            assert np.array_equal(greedy_count(unpack_windows(p, False), 8, 4 ** 8), want)
            assert np.array_equal(p.bg_counts, po.bg_counts(codes, offs, 2))
        else:
            assert cur == ref, nt


@pytest.mark.parametrize("W,M", [(4, 64), (8, 64), (10, 256), (14, 100)])
def test_packer_one_pass_path_equals_the_general_path(monkeypatch, W, M):
    """inputs made of whole runs only (no invalid base, every L >= W) take the one-pass packer: its stream, items
    and counters must be those of the general two-pass packer, for any thread count and ragged lengths"""
    rng = np.random.default_rng(100 + W)
    lens = rng.integers(W, 700, size=4000)
    lens[:50] = W  # exactly one window
    lens[50:60] = rng.integers(2000, 5000, size=10)  # several items per sequence
    codes = rng.integers(1, 5, size=int(lens.sum())).astype(np.uint8)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)

    def snapshot():
        p = pk.Packed(codes, offs, W, M)
        return (p.words.tobytes(), p.items.tobytes(), p.bg_counts.tobytes(), p.n_windows, p.n_bases, p.max_bin_bound, p.all_whole,
                p.max_len, p.n_sequences)

    monkeypatch.setenv("PENGK_PACK_GENERAL", "1")
    monkeypatch.setenv("PENGK_PACK_THREADS", "3")
    want = snapshot()
    assert want[6] == 1
    monkeypatch.setenv("PENGK_PACK_GENERAL", "0")
    for nt in (1, 2, 5, 16):
        monkeypatch.setenv("PENGK_PACK_THREADS", str(nt))
        assert snapshot() == want, nt
    assert np.array_equal(pk.Packed(codes, offs, W, M).bg_counts, po.bg_counts(codes, offs, 2))


def test_packer_one_pass_path_gives_way_on_any_irregular_sequence(monkeypatch):
    """one invalid base (anywhere, incl. the last byte) or one sequence shorter than W sends the whole input down the
    general path; the result must not depend on which path was tried first"""
    rng = np.random.default_rng(77)
    W = 8
    lens = rng.integers(W, 300, size=500)
    base = rng.integers(1, 5, size=int(lens.sum())).astype(np.uint8)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    for where, val in ((0, 0), (len(base) - 1, 0), (len(base) // 2, 7), (int(offs[250]) + 3, 200), (int(offs[100 + 1]) - 1, 0)):
        codes = base.copy()
        codes[where] = val
        monkeypatch.setenv("PENGK_PACK_GENERAL", "1")
        a = pk.Packed(codes, offs, W, 64)
        monkeypatch.setenv("PENGK_PACK_GENERAL", "0")
        b = pk.Packed(codes, offs, W, 64)
        assert a.all_whole == 0 and b.all_whole == 0
        assert a.words.tobytes() == b.words.tobytes() and a.items.tobytes() == b.items.tobytes()
        assert np.array_equal(a.bg_counts, b.bg_counts) and a.n_windows == b.n_windows
        if val == 0:  # the reader only produces codes 0..4; larger bytes are merely 'invalid' to the packer
            assert np.array_equal(b.bg_counts, po.bg_counts(codes, offs, 2))
    lens2 = lens.copy()
    lens2[123] = W - 1
    offs2 = np.concatenate([[0], np.cumsum(lens2)]).astype(np.int64)
    codes2 = base[:int(lens2.sum())]
    monkeypatch.setenv("PENGK_PACK_GENERAL", "1")
    a = pk.Packed(codes2, offs2, W, 64)
    monkeypatch.setenv("PENGK_PACK_GENERAL", "0")
    b = pk.Packed(codes2, offs2, W, 64)
    assert a.all_whole == 0 and b.all_whole == 0 and a.words.tobytes() == b.words.tobytes() and a.items.tobytes() == b.items.tobytes()


class PackTarget(pk.C.Structure):
    _fields_ = [("words", pk.C.POINTER(pk.C.c_uint64)), ("words_cap", pk.C.c_uint64), ("items", pk.C.POINTER(pk.C.c_uint64)),
                ("items_cap", pk.C.c_uint64), ("word_cursor", pk.C.c_uint64), ("item_cursor", pk.C.c_uint64)]


@pytest.mark.parametrize("irregular", [False, True])
def test_pack_append_collects_chunks_like_separate_packs(irregular):
    """pengk_pack_append (the CLI's streaming ingest: chunks of the input packed from many threads into ONE pair of
    buffers) -- every chunk's region holds what pengk_pack makes of the chunk alone, its items carry absolute stream
    offsets, the figures (windows, bounds, background counters) are the chunk's; cursors advance atomically."""
    import threading
    C = pk.C
    W, L, n_chunks, per = 10, 150, 12, 700
    rng = np.random.default_rng(3)
    chunks = []
    for c in range(n_chunks):
        codes = rng.integers(1, 5, size=per * L).astype(np.uint8)
        offs = np.arange(per + 1, dtype=np.int64) * L
        if irregular and c % 3 == 1:  # invalid bases / a short record: this chunk takes the general path
            codes[rng.integers(0, codes.size, 40)] = 0
        chunks.append((codes, offs))
    words = np.zeros(n_chunks * (per * L // 32 + 8) + 16, np.uint64)
    items = np.zeros(n_chunks * per * 2 + 16, np.uint64)
    tg = PackTarget(words.ctypes.data_as(C.POINTER(C.c_uint64)), words.size, items.ctypes.data_as(C.POINTER(C.c_uint64)), items.size, 0, 0)
    res = [None] * n_chunks

    def work(c):
        st = pk.PackedStruct()
        rc = pk.lib().pengk_pack_append(chunks[c][0].ctypes.data, chunks[c][1].ctypes.data, per, W, 0, C.byref(tg), C.byref(st))
        res[c] = (rc, st)

    th = [threading.Thread(target=work, args=(c,)) for c in range(n_chunks)]
    [t.start() for t in th]
    [t.join() for t in th]
    used_w, used_i = 0, 0
    for c, (rc, st) in enumerate(res):
        assert rc == 0
        alone = pk.Packed(chunks[c][0], chunks[c][1], W)
        w0 = (C.addressof(st.words.contents) - words.ctypes.data) // 8
        i0 = (C.addressof(st.items.contents) - items.ctypes.data) // 8 if st.n_items else 0
        assert np.array_equal(words[w0:w0 + len(alone.words)], alone.words) and not words[w0 + len(alone.words):w0 + st.n_words].any()
        assert st.n_items == len(alone.items)
        assert np.array_equal(items[i0:i0 + st.n_items], alone.items + np.uint64(32 * w0))
        assert (st.n_windows, st.max_bin_bound, st.all_whole) == (alone.n_windows, alone.max_bin_bound, alone.all_whole)
        assert list(st.bg_counts) == alone.bg_counts.tolist()
        used_w += st.n_words
        used_i += st.n_items
    assert tg.word_cursor == used_w and tg.item_cursor == used_i
    # a full buffer is an error, not an overrun
    small = PackTarget(words.ctypes.data_as(C.POINTER(C.c_uint64)), 8, items.ctypes.data_as(C.POINTER(C.c_uint64)), items.size, 0, 0)
    st = pk.PackedStruct()
    assert pk.lib().pengk_pack_append(chunks[0][0].ctypes.data, chunks[0][1].ctypes.data, per, W, 0, C.byref(small), C.byref(st)) == pk.ERR_RANGE


def test_the_committed_traffic_profile_is_not_older_than_the_count_kernels():
    """bench.py copies `roofline.traffic` / `roofline_issue` from profiles/traffic_by_config.json (rocprofv3 --pmc passes
    cannot run inside the driver's bench), every entry stamped with the commit it was measured at.  A change of the count
    kernels must not ship with a stale figure: csrc/count.hip as it stands must be the file of every entry's stamp, byte for
    byte (an experiment that was committed and reverted leaves it so) -- otherwise tools/profile_round.sh has to be run
    again.  (No git history on the GPU box: the test runs where the repository is.)"""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.isdir(os.path.join(root, ".git")):
        pytest.skip("no git history here")

    def git(*a):
        return subprocess.run(["git", "-C", root] + list(a), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    prof = json.load(open(os.path.join(root, "profiles", "traffic_by_config.json")))
    assert prof, "profiles/traffic_by_config.json is empty"
    for key, entry in prof.items():
        stamp = entry.get("commit")
        assert stamp and git("cat-file", "-e", stamp + "^{commit}").returncode == 0, (key, stamp)
        src = "peng-motif_amd/csrc/stats.hip" if key.startswith("sweep_") else "peng-motif_amd/csrc/count.hip"  # (K2+K3's entries: the sweep kernel)
        r = git("diff", "--quiet", stamp, "--", src)  # (the working tree against the stamp)
        assert r.returncode == 0, ("profiles/traffic_by_config.json[%s] was measured at %s and %s has changed since: "
                                   "run tools/profile_round.sh again" % (key, stamp, src))
