"""End-to-end: this repository's peng_motif (C++ host mirror over libpengk.so, GPU) against the
reference CLI's MEME / JSON / stdout for the same inputs (tests/golden/cli, produced by
tests/golden/make_cli_golden.py from the compiled reference).

Bar: same motifs in the same order, integer fields (w, nsites, opt_bg_order) and the motif names
identical; header floats equal as printed; PWM probabilities within 1e-4 absolute where the EM ran (the
reference's own float32 accumulation error) and as printed (8 decimals) where it did not;
the stdout trace up to the EM stage identical line by line."""
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
         "cli_torture_w6", "cli_mafk100_w8_bg1_enrich", "cli_mafk_w10_bg0_nofilter"]


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


def run_cli(tmp_path, name):
    args = open(os.path.join(GOLD, "cli", name + ".args")).read().split()
    meme, js = str(tmp_path / "out.meme"), str(tmp_path / "out.json")
    r = subprocess.run([CLI, os.path.join(GOLD, args[0])] + args[1:] + ["-o", meme, "-j", js], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return meme, js, r.stdout.decode()


@pytest.mark.parametrize("name", CASES)
def test_cli_matches_reference(tmp_path, name):
    assert os.path.exists(CLI), "build the host mirror: make -C peng-motif_amd/host"
    meme, js, stdout = run_cli(tmp_path, name)
    em_ran = "--no-em" not in open(os.path.join(GOLD, "cli", name + ".args")).read()
    bg, got = parse_meme(meme)
    bg_ref, want = parse_meme(os.path.join(GOLD, "cli", name + ".meme"))
    assert bg == bg_ref
    assert [m[0] for m in got] == [m[0] for m in want]
    for (n1, h1, p1), (n2, h2, p2) in zip(got, want):
        for key in ("alength", "w", "nsites", "opt_bg_order", "bg_prob"):
            assert h1[key] == h2[key], (n1, key)
        assert abs(float(h1["log(Pval)"]) - float(h2["log(Pval)"])) <= 1e-3 * max(1.0, abs(float(h2["log(Pval)"])))
        assert p1.shape == p2.shape
        assert np.abs(p1 - p2).max() <= (1e-4 if em_ran else 2e-8), n1
    # JSON: same schema and values
    a, b = json.load(open(js)), json.load(open(os.path.join(GOLD, "cli", name + ".json")))
    assert a["alphabet"] == b["alphabet"] and a["bg"] == b["bg"] and len(a["patterns"]) == len(b["patterns"])
    for x, y in zip(a["patterns"], b["patterns"]):
        for key in ("iupac_motif", "pattern_length", "sites", "opt_bg_order", "bg_prob"):
            assert x[key] == y[key]
        assert np.abs(np.array(x["pwm"]) - np.array(y["pwm"])).max() <= (1e-4 if em_ran else 3e-8)
    # stdout: everything before the EM stage is a deterministic function of bit-identical tables
    ref_out = open(os.path.join(GOLD, "cli", name + ".stdout")).read()
    cut = "[STATUS] Optimizing expectation-maximization"
    assert stdout.split(cut)[0] == ref_out.split(cut)[0]
    # ... and after it the same lines modulo the last printed digit of the information content
    tail_a, tail_b = stdout.split(cut)[1].split("\n"), ref_out.split(cut)[1].split("\n")
    assert [re.sub(r"avg. info: [0-9.]+", "", l) for l in tail_a] == [re.sub(r"avg. info: [0-9.]+", "", l) for l in tail_b]


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
