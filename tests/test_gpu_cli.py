"""End-to-end: this repository's peng_motif (C++ host mirror over libpengk.so, GPU) against the
reference CLI's MEME / JSON / stdout for the same inputs (tests/golden/cli, produced by
tests/golden/make_cli_golden.py from the compiled reference).

Bar: MEME, JSON and stdout identical byte for byte (the CLI's default EM mode reproduces the reference's float32
summation order); with the throughput EM modes (PENGK_EM_FAST=0/1) the documented tolerances instead."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "peng-motif_amd", "host")
CLI = os.path.join(HOST, "peng_motif")
SELFTEST = os.path.join(HOST, "host_selftest")
GOLD = os.path.join(ROOT, "tests", "golden")

CASES = ["cli_mafk100_w8", "cli_mafk100_w6_plus_noem", "cli_mafk100_w8_logpval_nomerge", "cli_mafk_w10", "cli_mafk_w10_plus",
         "cli_torture_w6", "cli_mafk100_w8_bg1_enrich", "cli_mafk_w10_bg0_nofilter", "cli_mafk_w2_bg0", "cli_mafk100_w2",
         # the command line of the reference's Python wrapper, with the wrapper's defaults (tests/golden/make_cli_golden.py)
         "cli_wrapper_mafk100_w6", "cli_wrapper_mafk_w10"]


def parse_meme(path):
    txt = open(path).read()
    bg = re.search(r"Background letter frequencies\n(.*)\n", txt).group(1)
    motifs = []
    for block in txt.split("MOTIF ")[1:]:
        lines = block.strip().split("\n")
        hdr = dict(re.findall(r"(\S+)= (\S+)", lines[1]))
        pwm = np.array([[float(x) for x in l.split()] for l in lines[2:] if l.strip()])
        motifs.append((lines[0].strip(), hdr, pwm))
    return bg, motifs


def run_cli(tmp_path, name, env=None):
    args = open(os.path.join(GOLD, "cli", name + ".args")).read().split()
    meme, js = str(tmp_path / "out.meme"), str(tmp_path / "out.json")
    r = subprocess.run([CLI, os.path.join(GOLD, args[0])] + args[1:] + ["-o", meme, "-j", js], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return meme, js, r.stdout.decode()


def compare_outputs(meme, js, stdout, ref_meme, ref_js, ref_out, em_ran):
    """The CLI runs the EM in the library's serial (bit-exact) mode, every table it reads is bit-exact and the host
    arithmetic is the reference's: the three outputs must be the reference's byte for byte."""
    assert open(meme).read() == open(ref_meme).read()
    assert open(js).read() == open(ref_js).read()
    assert stdout == ref_out


def compare_outputs_within_tolerance(meme, js, stdout, ref_meme, ref_js, ref_out, em_ran):
    """For the throughput EM modes (PENGK_EM_FAST=0/1: fp64 tree sums): same motifs in the same order, integer fields and
    printed header floats identical, PWMs within 1e-4 absolute where the EM ran (the reference's own float32 accumulation
    error), the stdout trace identical up to the EM stage."""
    bg, got = parse_meme(meme)
    bg_ref, want = parse_meme(ref_meme)
    assert bg == bg_ref
    assert [m[0] for m in got] == [m[0] for m in want]
    for (n1, h1, p1), (n2, h2, p2) in zip(got, want):
        for key in ("alength", "w", "nsites", "opt_bg_order", "bg_prob"):
            assert h1[key] == h2[key], (n1, key)
        assert abs(float(h1["log(Pval)"]) - float(h2["log(Pval)"])) <= 1e-3 * max(1.0, abs(float(h2["log(Pval)"])))
        assert p1.shape == p2.shape
        assert np.abs(p1 - p2).max() <= (1e-4 if em_ran else 2e-8), n1
    a, b = json.load(open(js)), json.load(open(ref_js))
    assert a["alphabet"] == b["alphabet"] and a["bg"] == b["bg"] and len(a["patterns"]) == len(b["patterns"])
    for x, y in zip(a["patterns"], b["patterns"]):
        for key in ("iupac_motif", "pattern_length", "sites", "opt_bg_order", "bg_prob"):
            assert x[key] == y[key]
        assert np.abs(np.array(x["pwm"]) - np.array(y["pwm"])).max() <= (1e-4 if em_ran else 3e-8)
    cut = "[STATUS] Optimizing expectation-maximization"
    assert stdout.split(cut)[0] == ref_out.split(cut)[0]
    tail_a, tail_b = stdout.split(cut)[1].split("\n"), ref_out.split(cut)[1].split("\n")
    assert [re.sub(r"avg. info: [0-9.]+", "", l) for l in tail_a] == [re.sub(r"avg. info: [0-9.]+", "", l) for l in tail_b]


@pytest.mark.parametrize("name", CASES)
def test_cli_matches_reference(tmp_path, name):
    assert os.path.exists(CLI), "build the host mirror: make -C peng-motif_amd/host"
    meme, js, stdout = run_cli(tmp_path, name)
    em_ran = "--no-em" not in open(os.path.join(GOLD, "cli", name + ".args")).read()
    compare_outputs(meme, js, stdout, os.path.join(GOLD, "cli", name + ".meme"), os.path.join(GOLD, "cli", name + ".json"),
                    open(os.path.join(GOLD, "cli", name + ".stdout")).read(), em_ran)


def test_cli_json_is_what_the_reference_wrapper_reads(tmp_path):
    """The reference's Python wrapper (scripts/shoot_peng.py, out of scope) is bound to the binary by two things only: the
    flag list it passes (:122-153 -- the cli_wrapper_* cases above run exactly that) and the JSON it loads and rewrites
    (:197-200, 236-250, 260-295): top-level alphabet / alphabet_length / bg / patterns, per pattern iupac_motif,
    pattern_length, sites, bg_prob, opt_bg_order, log(Pval), pwm (pattern_length rows of alphabet_length numbers).  A user
    who points the wrapper's PENG at this binary finds every key with the type the wrapper's writers format."""
    meme, js, stdout = run_cli(tmp_path, "cli_wrapper_mafk_w10")
    d = json.load(open(js))
    assert set(d) == {"alphabet", "alphabet_length", "bg", "patterns"}
    assert d["alphabet"] == "ACGT" and d["alphabet_length"] == 4 and len(d["bg"]) == 4 and len(d["patterns"]) > 0
    for p in d["patterns"]:
        assert set(p) == {"iupac_motif", "pattern_length", "sites", "log(Pval)", "bg_prob", "opt_bg_order", "pwm"}
        assert isinstance(p["iupac_motif"], str) and len(p["iupac_motif"]) == p["pattern_length"]
        assert isinstance(p["sites"], int) and isinstance(p["opt_bg_order"], int)
        assert isinstance(p["log(Pval)"], (int, float)) and isinstance(p["bg_prob"], (int, float))  # (an underflown bg_prob prints as 0, as the reference's does)
        assert len(p["pwm"]) == p["pattern_length"] and all(len(row) == 4 for row in p["pwm"])
        assert all(abs(sum(row) - 1.0) < 1e-5 for row in p["pwm"])


def test_cli_host_copies_released_above_a_resident_size(tmp_path):
    """The packed words / items the reader builds on the host stay mapped after the upload for small inputs (unmapping
    0.6 GB costs the bench run ~35 ms of TLB shoot-downs) and go back for large ones: PENGK_HOST_RELEASE_MB (default
    1024) is compared with what is RESIDENT -- the words and items written --, not with the worst-case reservation the file
    size allows (that comparison released the bench's 0.58 GB behind a 2 GiB reservation: 0.23 s instead of 0.17 s end to
    end, round 5).  Forced release (= 1) and no release (= 0): the reference's outputs either way."""
    name = "cli_mafk_w10"
    ref = [os.path.join(GOLD, "cli", name + e) for e in (".meme", ".json")]
    for mb, expect in (("1", True), ("0", False), (None, False)):
        env = dict(os.environ, PENGK_TIMING="1")
        if mb is not None:
            env["PENGK_HOST_RELEASE_MB"] = mb
        args = open(os.path.join(GOLD, "cli", name + ".args")).read().split()
        meme, js = str(tmp_path / "o.meme"), str(tmp_path / "o.json")
        r = subprocess.run([CLI, os.path.join(GOLD, args[0])] + args[1:] + ["-o", meme, "-j", js], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           timeout=600, env=env)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        assert (b"host copies of the packed input released" in r.stderr) == expect, (mb, r.stderr.decode()[-1500:])
        assert open(meme).read() == open(ref[0]).read() and open(js).read() == open(ref[1]).read()
        assert r.stdout.decode() == open(os.path.join(GOLD, "cli", name + ".stdout")).read()


@pytest.mark.parametrize("mode", ["0", "1"])
@pytest.mark.parametrize("name", ["cli_mafk100_w8", "cli_mafk_w10", "cli_mafk_w10_plus"])
def test_cli_throughput_em_modes_within_tolerance(tmp_path, name, mode):
    meme, js, stdout = run_cli(tmp_path, name, env=dict(os.environ, PENGK_EM_FAST=mode))
    compare_outputs_within_tolerance(meme, js, stdout, os.path.join(GOLD, "cli", name + ".meme"),
                                     os.path.join(GOLD, "cli", name + ".json"),
                                     open(os.path.join(GOLD, "cli", name + ".stdout")).read(), True)


REF_CLI = os.path.join(ROOT, "oracle", "_ref", "peng_motif_ref")
RANDOM_CASES = [(0, ["-w", "8"]), (1, ["-w", "10"]), (2, ["-w", "6", "--strand", "PLUS"]), (3, ["-w", "10", "--strand", "PLUS", "--no-merging"]),
                (4, ["-w", "8", "--no-em", "-t", "5"]), (5, ["-w", "10", "--optimization_score", "LOGPVAL"]),
                (6, ["-w", "8", "--no-neighbor-filtering", "--max-optimized-patterns", "20"]), (7, ["-w", "12", "--count-threshold", "2"])]


@pytest.mark.parametrize("seed,flags", RANDOM_CASES)
def test_cli_random_inputs_against_reference_binary(tmp_path, seed, flags):
    """The compiled reference travels with the repository when it was built (oracle/_ref/, build container only): run
    BOTH programs on a random FASTA with planted, mutated motifs (some N, lower case, ragged lengths) and compare their
    outputs like the goldens -- seed ranking, lockstep hill-climb, PWMs, EM and merging on inputs nobody looked at."""
    if not os.path.exists(REF_CLI):
        pytest.skip("oracle/_ref/peng_motif_ref not present (the reference is only built in the build container)")
    rng = np.random.default_rng(500 + seed)
    n, L = int(rng.integers(300, 1500)), int(rng.integers(60, 220))
    motifs = ["".join(rng.choice(list("ACGT"), size=int(rng.integers(8, 13)))) for _ in range(3)]
    lines = []
    for i in range(n):
        s = rng.choice(list("ACGT"), size=int(L + rng.integers(-20, 21)))
        for m in motifs:
            if rng.random() < 0.35:
                mm = list(m)
                if rng.random() < 0.4:
                    mm[int(rng.integers(0, len(mm)))] = str(rng.choice(list("ACGT")))
                at = int(rng.integers(0, len(s) - len(mm)))
                s[at:at + len(mm)] = mm
        if rng.random() < 0.05:
            s[int(rng.integers(0, len(s)))] = "N"
        t = "".join(s)
        if rng.random() < 0.1:
            t = t.lower()
        lines.append(">r%d\n%s\n" % (i, t))
    fa = tmp_path / "random.fa"
    fa.write_text("".join(lines))
    outs = {}
    for tag, exe in (("ref", REF_CLI), ("here", CLI)):
        meme, js = str(tmp_path / (tag + ".meme")), str(tmp_path / (tag + ".json"))
        r = subprocess.run([exe, str(fa)] + flags + ["-o", meme, "-j", js], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, (tag, r.stderr.decode()[-2000:])
        outs[tag] = (meme, js, r.stdout.decode())
    if not os.path.exists(outs["ref"][0]) or "MOTIF" not in open(outs["ref"][0]).read():
        assert outs["here"][2] == outs["ref"][2]  # nothing found: the traces must still agree
        return
    compare_outputs(*outs["here"], outs["ref"][0], outs["ref"][1], outs["ref"][2], "--no-em" not in flags)


@pytest.mark.parametrize("flags", [["-w", "8"], ["-w", "10", "--strand", "PLUS"]])
def test_cli_genome_scale_records_against_reference_binary(tmp_path, flags):
    """Records far longer than any item or reader chunk: one of 3 Mbp written as ONE line, one of 1.2 Mbp in 60-column
    lines with a poly-A run of 40 kbp and a stretch of N inside, one shorter than the pattern, a few hundred ordinary ones
    with a planted motif -- the sharded reader, the item splitter (a record becomes thousands of items whose non-overlap
    state crosses item borders) and the deferred fix-up against the compiled reference, byte for byte."""
    if not os.path.exists(REF_CLI):
        pytest.skip("oracle/_ref/peng_motif_ref not present (the reference is only built in the build container)")
    rng = np.random.default_rng(77)
    acgt = np.array(list("ACGT"))
    motif = "GCTGAGTCAT"

    def plant(a, every):
        for at in range(1000, len(a) - 20, every):
            a[at:at + len(motif)] = list(motif)
        return a
    big = plant(acgt[rng.integers(0, 4, 3_000_000)], 9973)
    mid = plant(acgt[rng.integers(0, 4, 1_200_000)], 7919)
    mid[300_000:340_000] = "A"
    mid[500_000:500_700] = "N"
    with open(tmp_path / "g.fa", "w") as f:
        f.write(">chrBig\n" + "".join(big) + "\n")
        f.write(">chrMid\n")
        m = "".join(mid)
        f.write("\n".join(m[i:i + 60] for i in range(0, len(m), 60)) + "\n")
        f.write(">tiny\nACGTA\n")
        for i in range(300):
            s = acgt[rng.integers(0, 4, int(rng.integers(80, 400)))]
            if i % 3 == 0:
                s[20:20 + len(motif)] = list(motif)
            f.write(">r%d\n%s\n" % (i, "".join(s)))
    outs = {}
    for tag, exe in (("ref", REF_CLI), ("here", CLI)):
        meme, js = str(tmp_path / (tag + ".meme")), str(tmp_path / (tag + ".json"))
        r = subprocess.run([exe, str(tmp_path / "g.fa")] + flags + ["-o", meme, "-j", js], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        assert r.returncode == 0, (tag, r.stderr.decode()[-2000:])
        outs[tag] = (meme, js, r.stdout.decode())
    assert "MOTIF" in open(outs["ref"][0]).read()
    compare_outputs(*outs["here"], outs["ref"][0], outs["ref"][1], outs["ref"][2], True)


@pytest.mark.parametrize("with_bg", [False, True])
def test_cli_stderr_warnings_match_the_reference(tmp_path, with_bg):
    """stderr: the reference warns about records without sequence as it meets them and about undefined bases only in the
    LAST record of a file (src/shared/SequenceSet.cpp:395-405), once per SequenceSet -- i.e. twice for the input file when
    no --background-sequences is given, because it reads that file a second time (src/Global.cpp:66-75).  Same lines, same
    order here."""
    if not os.path.exists(REF_CLI):
        pytest.skip("oracle/_ref/peng_motif_ref not present (the reference is only built in the build container)")
    rng = np.random.default_rng(77)
    recs = []
    for i in range(400):
        s = "".join(rng.choice(list("ACGT"), size=80))
        if i % 3 == 0:
            s = s[:30] + "GCTGAGTCAT" + s[40:]
        if i == 17:
            s = s[:10] + "N" + s[11:]          # an undefined base in the middle of the file: silent in the reference
        recs.append(">r%d\n%s\n" % (i, s))
    recs.insert(100, ">empty\n")               # a record without sequence
    recs.append(">last\nACGTNACGTXACGT" + "ACGT" * 10 + "\n")  # undefined bases in the last record
    fa = tmp_path / "warn.fa"
    fa.write_text("".join(recs))
    flags = ["-w", "8"]
    if with_bg:
        bg = tmp_path / "bg.fa"
        bg.write_text("".join(">b%d\n%s\n" % (i, "".join(rng.choice(list("ACGT"), size=100))) for i in range(200)) + ">blast\nACGTRACGT" + "A" * 30 + "\n")
        flags += ["--background-sequences", str(bg)]
    warn = {}
    for tag, exe in (("ref", REF_CLI), ("here", CLI)):
        r = subprocess.run([exe, str(fa)] + flags + ["-o", str(tmp_path / (tag + ".meme"))], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, (tag, r.stderr.decode()[-2000:])
        warn[tag] = [l for l in r.stderr.decode().split("\n") if l.startswith("Warning:")]
    assert len(warn["ref"]) >= 3
    assert warn["here"] == warn["ref"]


def test_host_selftest_reference_unit_checks():
    """the checks of the reference's gtest fixture (test/test_base_pattern.cpp) on this repository's classes,
    on the reference fixture's own 3-record input"""
    fa = os.path.join(GOLD, "default_sequence_set.fa")
    r = subprocess.run([SELFTEST, fa], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr).decode()[-2000:]
    assert b"host_selftest ok" in r.stdout


def test_cli_exit_codes(tmp_path):
    fa = os.path.join(GOLD, "MafK_100seqs.fasta")
    assert subprocess.run([CLI, "-h"], stdout=subprocess.DEVNULL).returncode == 0
    assert subprocess.run([CLI, fa, "-w", "7"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode == 4
    assert subprocess.run([CLI, fa, "-w"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode == 4
    assert subprocess.run([CLI], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode == 255  # exit(-1)
    bad = tmp_path / "bad.fa"
    bad.write_text(">a\nAC GT\n")
    assert subprocess.run([CLI, str(bad)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode == 1


def test_device_seed_selection_is_the_reference_set_up_to_strand(tmp_path):
    """PENGK_SEED_SELECT=device (opt-in): the printed seed table holds the same k-mers as the default
    (reference-ranked) run up to the strand a reverse-complement pair is named on, with the same counts and z-scores."""
    comp = str.maketrans("ACGT", "TGCA")

    def seed_table(env_extra):
        r = subprocess.run([CLI, os.path.join(ROOT, "tests", "golden", "MafK.fasta"), "-w", "10", "--no-em", "--no-merging"], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, env=dict(os.environ, **env_extra), timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        lines = r.stdout.decode().split("\n")
        start = next(i for i, l in enumerate(lines) if "pattern" in l and "zscore" in l) + 2
        rows = []
        for l in lines[start:]:
            f = l.split()
            if len(f) != 4:
                break
            rows.append((min(f[0], f[0].translate(comp)[::-1]), f[1], f[3]))
        return rows

    ref = seed_table({})
    dev = seed_table({"PENGK_SEED_SELECT": "device"})
    assert len(ref) > 100 and sorted(ref) == sorted(dev)


def test_merging_through_the_device_similarity_grid_prints_what_the_exact_path_prints(tmp_path):
    """Many motifs (z threshold 4, up to 1000 optimised patterns: ~220 PWMs reach the merge loop): the device similarity
    grid picks the candidate pairs, the reference's arithmetic decides among them -- stdout (every `merge:` line), MEME
    and JSON equal the exact-scores-only path (PENGK_MERGE_GRID=host) byte for byte."""
    outs = []
    for tag, extra in (("grid", {}), ("host", {"PENGK_MERGE_GRID": "host"})):
        meme, js = tmp_path / (tag + ".meme"), tmp_path / (tag + ".json")
        r = subprocess.run([CLI, os.path.join(ROOT, "tests", "golden", "MafK.fasta"), "-w", "10", "-t", "4", "--max-optimized-patterns",
                            "1000", "--minimum-processed-patterns", "1000", "-o", str(meme), "-j", str(js)], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, env=dict(os.environ, **extra), timeout=900)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        outs.append((r.stdout, meme.read_bytes(), js.read_bytes()))
    assert outs[0][0].count(b"\nmerge: ") >= 20 and outs[0][0].count(b"\nem: ") >= 150
    assert outs[0] == outs[1]


def test_pattern_lengths_beyond_the_tables_end_like_the_references_too_long(tmp_path):
    """W = 16 passes the reference's own limit (11^16 < 2^64, /root/reference/src/peng.cpp:325-330) and then needs 4^16-entry
    tables (~150 GB there; its README advises W <= 12).  This build's tables end at W = 14: the run ends with the two
    lines and the exit code the reference gives a pattern length beyond its limit -- after reading the input, like the
    reference -- and W = 18, beyond both limits, prints the reference's own numbers."""
    cli = os.path.join(HOST, "peng_motif")
    fa = os.path.join(GOLD, "MafK_100seqs.fasta")
    r = subprocess.run([cli, fa, "-w", "16"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 1 and b"Warning: pattern length too long!" in r.stderr and b"max pattern length: 14" in r.stderr
    r = subprocess.run([cli, fa, "-w", "18"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 1 and b"Warning: pattern length too long!" in r.stderr and b"max pattern length: 31" in r.stderr


@pytest.mark.parametrize("flags", [["-w", "2"], ["-w", "2", "--bg-model-order", "0", "-t", "2", "--count-threshold", "1"],
                                   ["-w", "2", "--strand", "PLUS", "--bg-model-order", "0", "-t", "1", "--no-em"]],
                         ids=["w2", "w2_bg0", "w2_plus_bg0_noem"])
@pytest.mark.parametrize("text", [b"", b">a\n>b\n", b">a\nA\n>b\nAC\n", b">a\nNNNNNNNNNNNN\n>b\nNANANANANA\n",
                                  b">a\nACGTACGTTTGACCA\n>b\nacgtnACGTNNacgggt\n>c\nTTTTTTTTTTTTTTTTTTTTTTTTT\n"],
                         ids=["empty", "headers_only", "too_short", "mostly_n", "tiny"])
def test_w2_on_degenerate_inputs_against_reference_binary(tmp_path, text, flags):
    """W = 2, the shortest pattern length the reference accepts (/root/reference/src/Global.cpp:103-106), on inputs without
    records, without a window, with more N than bases: exit code, stdout and MEME of both programs byte for byte."""
    if not os.path.exists(REF_CLI):
        pytest.skip("oracle/_ref/peng_motif_ref not present (the reference is only built in the build container)")
    fa = tmp_path / "d.fa"
    fa.write_bytes(text)
    outs = []
    for tag, exe in (("ref", REF_CLI), ("here", CLI)):
        meme = tmp_path / (tag + ".meme")
        r = subprocess.run([exe, str(fa)] + flags + ["-o", str(meme)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        outs.append((r.returncode, r.stdout, meme.read_bytes() if meme.exists() else None))
    assert outs[0] == outs[1], (outs[0][0], outs[1][0], outs[0][1][-300:], outs[1][1][-300:])
