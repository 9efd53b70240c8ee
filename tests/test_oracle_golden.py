"""The oracle (oracle/peng_oracle.cpp) against the golden vectors dumped from the compiled
reference (tests/golden/make_golden.py).  Bit-exact: integers equal, float32 bit patterns equal.
CPU only."""
import hashlib
import os

import numpy as np
import pytest

from oracle import oracle as po

CASES = ["mafk100_w8_both", "mafk100_w8_plus", "mafk100_w6_both", "torture_w6_both", "torture_w6_plus",
         "torture_w4_both", "torture_w8_plus", "mafk_w10_both", "mafk_w10_plus", "torture_w2_both", "mafk100_w2_plus"]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def bits_equal(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    return a.shape == b.shape and a.tobytes() == b.tobytes()


_cache = {}


def pipeline(golden_dir, name):
    """oracle end-to-end on a golden case -> dict of arrays (cached per session)."""
    if name in _cache:
        return _cache[name]
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    W, both, K = int(g["W"]), bool(g["both"]), int(g["K"])
    codes, offs = po.read_fasta(os.path.join(golden_dir, str(g["fasta"])))
    n = po.bg_counts(codes, offs, 2)
    V = po.bg_V(n, 2)
    counts, ltot = po.count(codes, offs, W, both)
    bgp = [po.bgprob(W, k, V, both) for k in range(K + 1)]
    e, lp, z = po.stats(W, counts, bgp[K], ltot)
    r = dict(g=g, W=W, both=both, K=K, codes=codes, offs=offs, bgcounts=n, V=V, counts=counts, ltot=ltot, bgp=bgp,
             expected=e, logp=lp, z=z)
    _cache[name] = r
    return r


@pytest.mark.parametrize("name", CASES)
def test_tables_bit_exact(golden_dir, name):
    r = pipeline(golden_dir, name)
    g = r["g"]
    assert len(r["offs"]) - 1 == int(g["N"])
    assert bits_equal(r["bgcounts"], g["bgcounts"])
    assert bits_equal(r["V"], g["V"])
    assert r["ltot"] == int(g["ltot"])
    arrays = dict(counts=r["counts"], expected=r["expected"], z=r["z"], logp=r["logp"])
    for k in range(r["K"] + 1):
        arrays["bgp%d" % k] = r["bgp"][k]
    if "counts" in g:
        if "codes" in g:
            assert bits_equal(r["codes"], g["codes"]) and bits_equal(r["offs"], g["offs"])
        for k, a in arrays.items():
            assert bits_equal(a, g[k]), k
    else:
        assert sha(r["codes"]) == str(g["sha_codes"]) and sha(r["offs"]) == str(g["sha_offs"])
        idx = g["slice_idx"].astype(np.int64)
        for k, a in arrays.items():
            assert bits_equal(a[idx], g["slice_" + k]), k
            assert sha(a) == str(g["sha_" + k]), k


@pytest.mark.parametrize("name", CASES)
def test_seed_selection(golden_dir, name):
    r = pipeline(golden_dir, name)
    g = r["g"]
    assert bits_equal(po.select(r["W"], r["z"], r["counts"], 10.0, 3, not r["both"], True), g["seeds"])
    assert bits_equal(po.select(r["W"], r["z"], r["counts"], 3.0, 1, not r["both"], False), g["seeds_nofilter_z3"])


@pytest.mark.parametrize("name", CASES)
def test_iupac_aggregation(golden_dir, name):
    r = pipeline(golden_dir, name)
    g = r["g"]
    N = int(g["N"])
    step = max(1, len(g["iupac_ids"]) // 300)  # keep the CPU suite short; make_golden.py checks all rows
    for i in range(0, len(g["iupac_ids"]), step):
        pid = int(g["iupac_ids"][i])
        st = po.iupac_aggregate(pid, r["W"], r["both"], r["counts"], r["bgp"][r["K"]], r["expected"])
        mi = po.mi_score(np.float32(st.sites), st.expected, N)
        got = np.array([st.bg_p, st.expected, st.zscore, st.log_pvalue, mi], np.float32).view(np.uint32)
        assert st.sites == int(g["iupac_sites"][i])
        assert po.iupac_count(pid, r["W"], r["both"], r["counts"]) == int(g["iupac_cc"][i])
        assert np.array_equal(got, g["iupac_fbits"][i]), po.iupac_str(pid, r["W"])
    for sid, bits in g["base_mi"]:
        got = np.array([po.mi_score(np.float32(r["counts"][int(sid)]), r["expected"][int(sid)], N)], np.float32)
        assert got.view(np.uint32)[0] == bits


@pytest.mark.parametrize("name", CASES)
def test_em(golden_dir, name):
    """mode 0 (serial float32) reproduces the reference bit for bit; mode 1 (fp64 accumulators)
    stays inside the envelope SURVEY.md A.7 measured for the reference's own rounding error."""
    r = pipeline(golden_dir, name)
    g = r["g"]
    for i in range(len(g["pwm_ids"])):
        p0, it, _ = po.em(r["W"], r["counts"], r["bgp"][r["K"]], g["pwm_pre"][i], 1e4, 0.08, 10, mode=0)
        assert it == int(g["em_iters"][i])
        assert bits_equal(p0, g["pwm_post"][i])
        p1, it1, _ = po.em(r["W"], r["counts"], r["bgp"][r["K"]], g["pwm_pre"][i], 1e4, 0.08, 10, mode=1)
        assert it1 == it
        assert np.abs(p1.astype(np.float64) - g["pwm_post"][i]).max() < 1e-4


def test_reference_unit_test_vectors():
    """The reference's own gtest pins (test/test_base_pattern.cpp:38-131): encodings only."""
    # check_reverse_complement / check_fast_reverse_complement (:52-68): W=4, "ACGT" is its own revcomp;
    # id("AAAC") = 1*4^3 -> revcomp "GTTT"
    acgt = 0 + 1 * 4 + 2 * 16 + 3 * 64
    assert po.revcomp(acgt, 4) == acgt
    aaac = 1 * 64
    gttt = 2 + 3 * 4 + 3 * 16 + 3 * 64
    assert po.revcomp(aaac, 4) == gttt
    for x in range(256):
        assert po.revcomp(po.revcomp(x, 4), 4) == x
    # iupac2base_patterns (:81-118): expansion of an IUPAC pattern has prod(|letter|) distinct members
    pid = po.iupac_id("ANSW")
    ids = po.iupac_expand(pid, 4, False)
    assert len(ids) == 1 * 4 * 2 * 2 and len(set(ids.tolist())) == 16
    for x in ids.tolist():
        s = po.kmer_str(x, 4)
        assert s[0] == "A" and s[2] in "CG" and s[3] in "AT"
    # emission order of the LIFO stack (src/iupac_pattern.cpp:386-405): last position fastest, rep0 then n-1..1
    assert [po.kmer_str(x, 2) for x in po.iupac_expand(po.iupac_id("NS"), 2, False).tolist()] == \
        ["AC", "AG", "TC", "TG", "GC", "GG", "CC", "CG"]


def test_fasta_quirks(tmp_path):
    """src/shared/SequenceSet.cpp:285-447: unterminated last line dropped, blank lines skipped,
    header without sequence skipped, non-ACGT -> 0, lower case accepted, space -> error."""
    p = tmp_path / "a.fa"
    p.write_bytes(b">a\nACgtN\n\n>b\n>c\nAC\nGT\n>d\nAAAA")
    codes, offs = po.read_fasta(str(p))
    assert offs.tolist() == [0, 5, 9]
    assert codes.tolist() == [1, 2, 3, 4, 0, 1, 2, 3, 4]
    p.write_bytes(b">a\nAC GT\n")
    with pytest.raises(ValueError):
        po.read_fasta(str(p))
    p.write_bytes(b"ACGT\n>a\nAC\n")
    with pytest.raises(ValueError):
        po.read_fasta(str(p))


def test_count_shard_additivity(golden_dir):
    """counts and ltot are exactly additive over whole-sequence shards."""
    codes, offs = po.read_fasta(os.path.join(golden_dir, "MafK_100seqs.fasta"))
    full, lt = po.count(codes, offs, 8, True)
    h = 11
    a, la = po.count(codes[:offs[h]], offs[:h + 1], 8, True)
    b, lb = po.count(codes[offs[h]:], offs[h:] - offs[h], 8, True)
    assert la + lb == lt and np.array_equal(a + b, full)


def test_synth_generator_properties():
    c, o = po.synth(1, 0, 2000, 200)
    assert c.min() == 1 and c.max() == 4 and len(o) == 2001
    # any shard can be generated independently
    c2, _ = po.synth(1, 500, 100, 200)
    assert np.array_equal(c2, c[500 * 200:600 * 200])
    planted = sum(bytes(c[i * 200:(i + 1) * 200] + 64).translate(bytes.maketrans(b"ABCD", b"ACGT")).count(b"GCTGAGTCAT") > 0
                  for i in range(2000))
    assert 150 < planted < 260
