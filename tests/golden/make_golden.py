#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (oracle/_ref/ref_dump) and pin the oracle.

Run only in the build container (needs /root/reference to build oracle/_ref):

    make -C oracle all ref && python tests/golden/make_golden.py

For every case it runs the reference-linked dumper, compares every array bit for bit with
oracle/peng_oracle.cpp, prints a report (exit 1 on any mismatch) and writes a compressed .npz
holding inputs and expected outputs.  W <= 8: full arrays.  W = 10: sha256 of each full array
plus slices (top-2000 by z, a strided sample) to keep the fixtures small.
"""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as po  # noqa: E402

REF_DUMP = os.path.join(ROOT, "oracle", "_ref", "ref_dump")

CASES = [
    # name, fasta, W, strand
    ("mafk100_w8_both", "MafK_100seqs.fasta", 8, "BOTH"),   # BASELINE config 1
    ("mafk100_w8_plus", "MafK_100seqs.fasta", 8, "PLUS"),
    ("mafk100_w6_both", "MafK_100seqs.fasta", 6, "BOTH"),
    ("torture_w6_both", "torture.fa", 6, "BOTH"),
    ("torture_w6_plus", "torture.fa", 6, "PLUS"),
    ("torture_w4_both", "torture.fa", 4, "BOTH"),
    ("torture_w8_plus", "torture.fa", 8, "PLUS"),
    ("mafk_w10_both", "MafK.fasta", 10, "BOTH"),            # BASELINE config 2
    ("mafk_w10_plus", "MafK.fasta", 10, "PLUS"),
    # W = 2 (round 4): the shortest pattern length the reference accepts (any even W, src/Global.cpp:103-106); its
    # background order is min(W - 1, 2) = 1, which models 2-mers exactly -- z ~ 0, no seeds, the tables are the test
    ("torture_w2_both", "torture.fa", 2, "BOTH"),
    ("mafk100_w2_plus", "MafK_100seqs.fasta", 2, "PLUS"),
]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def parse_hex_table(path, nint):
    rows = []
    extra = []
    for line in open(path):
        if line.startswith("#"):
            extra.append(line[1:].split())
            continue
        t = line.split()
        ints = [int(x) for x in t[:nint]]
        bits = np.array([int(x, 16) for x in t[nint:]], np.uint32)
        rows.append((ints, bits))
    return rows, extra


def run_case(name, fasta, W, strand):
    both = strand == "BOTH"
    tmp = tempfile.mkdtemp(prefix="refdump_")
    subprocess.check_call([REF_DUMP, os.path.join(HERE, fasta), str(W), strand, tmp], stderr=subprocess.DEVNULL)
    ld = lambda f, t: np.fromfile(os.path.join(tmp, f), t)  # noqa: E731
    meta = dict(l.split() for l in open(os.path.join(tmp, "meta.txt")))
    K = int(meta["K"])
    ref = dict(
        codes=ld("codes.u8", np.uint8), offs=ld("offs.i64", np.int64), bgcounts=ld("bgcounts.i32", np.int32).astype(np.int64),
        V=ld("V.f32", np.float32), counts=ld("counts.u64", np.uint64), expected=ld("expected.f32", np.float32),
        z=ld("z.f32", np.float32), logp=ld("logp.f32", np.float32), seeds=ld("seeds.u64", np.uint64),
        seeds_nofilter_z3=ld("seeds_nofilter_z3.u64", np.uint64), ltot=np.uint64(int(meta["ltot"])), N=np.int64(int(meta["N"])),
    )
    for k in range(K + 1):
        ref["bgp%d" % k] = ld("bgp%d.f32" % k, np.float32)

    # ---- pin the oracle -------------------------------------------------------------------------
    bad = []

    def chk(what, a, b):
        a = np.asarray(a)
        b = np.asarray(b)
        same = a.shape == b.shape and a.tobytes() == b.tobytes()
        if not same:
            bad.append(what)
        return same

    codes, offs = po.read_fasta(os.path.join(HERE, fasta))
    chk("fasta.codes", codes, ref["codes"])
    chk("fasta.offs", offs, ref["offs"])
    n = po.bg_counts(codes, offs, 2)
    chk("bgcounts", n, ref["bgcounts"])
    V = po.bg_V(n, 2)
    chk("V", V, ref["V"])
    counts, ltot = po.count(codes, offs, W, both)
    chk("counts", counts, ref["counts"])
    chk("ltot", np.uint64(ltot), ref["ltot"])
    bgp = [po.bgprob(W, k, V, both) for k in range(K + 1)]
    for k in range(K + 1):
        chk("bgp%d" % k, bgp[k], ref["bgp%d" % k])
    e, lp, z = po.stats(W, counts, bgp[K], ltot)
    chk("expected", e, ref["expected"])
    chk("z", z, ref["z"])
    chk("logp", lp, ref["logp"])
    chk("seeds", po.select(W, z, counts, 10.0, 3, not both, True), ref["seeds"])
    chk("seeds_nofilter_z3", po.select(W, z, counts, 3.0, 1, not both, False), ref["seeds_nofilter_z3"])

    # IUPAC aggregation table
    rows, extra = parse_hex_table(os.path.join(tmp, "iupac.txt"), 3)
    iu_ids = np.array([r[0][0] for r in rows], np.uint64)
    iu_sites = np.array([r[0][1] for r in rows], np.uint64)
    iu_cc = np.array([r[0][2] for r in rows], np.uint64)
    iu_f = np.array([r[1] for r in rows], np.uint32).reshape(len(rows), 5)  # bg_p expected z logp mi (float bits)
    nbad = 0
    for i, pid in enumerate(iu_ids):
        st = po.iupac_aggregate(int(pid), W, both, counts, bgp[K], e)
        mi = po.mi_score(np.float32(st.sites), st.expected, int(ref["N"]))
        got = np.array([st.bg_p, st.expected, st.zscore, st.log_pvalue, mi], np.float32).view(np.uint32)
        cc = po.iupac_count(int(pid), W, both, counts)
        if st.sites != iu_sites[i] or cc != iu_cc[i] or not np.array_equal(got, iu_f[i]):
            nbad += 1
            if nbad < 4:
                print("   iupac mismatch", po.iupac_str(int(pid), W), st.sites, iu_sites[i], cc, iu_cc[i], got, iu_f[i])
    if nbad:
        bad.append("iupac(%d/%d)" % (nbad, len(rows)))
    base_mi = np.array([[int(t[1]), int(t[2], 16)] for t in extra if t and t[0] == "basemi"], np.uint64).reshape(-1, 2)
    for sid, bits in base_mi:
        got = np.array([po.mi_score(np.float32(counts[int(sid)]), e[int(sid)], int(ref["N"]))], np.float32).view(np.uint32)[0]
        if got != bits:
            bad.append("base_mi")
            break

    # EM: pre-EM PWMs -> post-EM PWMs (reference's serial float32 sums == oracle mode 0)
    pre, _ = parse_hex_table(os.path.join(tmp, "pwm_noem.txt"), 2)
    post, _ = parse_hex_table(os.path.join(tmp, "pwm_em.txt"), 2)
    assert len(pre) == len(post)
    pwm_ids = np.array([r[0][0] for r in pre], np.uint64)
    pwm_sites = np.array([r[0][1] for r in pre], np.uint64)
    pwm_pre = np.array([r[1][3:] for r in pre], np.uint32).reshape(len(pre), W, 4).view(np.float32)
    pwm_post = np.array([r[1][3:] for r in post], np.uint32).reshape(len(post), W, 4).view(np.float32)
    pwm_meta = np.array([r[1][:3] for r in pre], np.uint32).reshape(len(pre), 3).view(np.float32)  # logp bg_p expected
    em_iters = np.zeros(len(pre), np.int32)
    em_dev64 = 0.0
    for i in range(len(pre)):
        assert pre[i][0][0] == post[i][0][0]
        p0, it, _ = po.em(W, counts, bgp[K], pwm_pre[i], 1e4, 0.08, 10, mode=0)
        em_iters[i] = it
        if p0.tobytes() != pwm_post[i].tobytes():
            bad.append("em[%d]" % i)
        p1, it1, _ = po.em(W, counts, bgp[K], pwm_pre[i], 1e4, 0.08, 10, mode=1)
        if it1 != it:
            bad.append("em_iters64[%d]" % i)
        em_dev64 = max(em_dev64, float(np.abs(p1.astype(np.float64) - pwm_post[i]).max()))

    print("%-18s W=%-2d %-4s N=%-5d ltot=%-8d seeds=%-4d iupac=%-4d pwms=%-3d ref-vs-fp64 EM max abs %.2e  %s" % (
        name, W, strand, int(ref["N"]), int(ref["ltot"]), len(ref["seeds"]), len(rows), len(pre), em_dev64,
        "OK" if not bad else "MISMATCH " + ",".join(bad)))

    # ---- write the fixture ------------------------------------------------------------------------
    g = dict(W=np.int32(W), both=np.int32(both), K=np.int32(K), fasta=np.array(fasta), N=ref["N"], ltot=ref["ltot"],
             bgcounts=ref["bgcounts"], V=ref["V"], seeds=ref["seeds"], seeds_nofilter_z3=ref["seeds_nofilter_z3"],
             iupac_ids=iu_ids, iupac_sites=iu_sites, iupac_cc=iu_cc, iupac_fbits=iu_f, base_mi=base_mi,
             pwm_ids=pwm_ids, pwm_sites=pwm_sites, pwm_pre=pwm_pre, pwm_post=pwm_post, pwm_meta=pwm_meta, em_iters=em_iters)
    big = ["counts", "expected", "z", "logp"] + ["bgp%d" % k for k in range(K + 1)]
    if W <= 8:
        for k in big:
            g[k] = ref[k]
        if len(ref["codes"]) < 100000:
            g["codes"] = ref["codes"]
            g["offs"] = ref["offs"]
    else:
        zf = ref["z"].astype(np.float64)
        zf[~np.isfinite(zf)] = -1e30
        top = np.argsort(-zf, kind="stable")[:2000].astype(np.uint64)
        stride = np.arange(0, 4 ** W, 997, dtype=np.uint64)
        idx = np.unique(np.concatenate([top, stride, ref["seeds"], ref["seeds_nofilter_z3"][:500]]))
        g["slice_idx"] = idx
        for k in big:
            g["sha_" + k] = np.array(sha(ref[k]))
            g["slice_" + k] = ref[k][idx.astype(np.int64)]
        g["sha_codes"] = np.array(sha(ref["codes"]))
        g["sha_offs"] = np.array(sha(ref["offs"]))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **g)
    return not bad


def main():
    if not os.path.exists(REF_DUMP):
        sys.exit("build the reference first: make -C oracle ref")
    ok = True
    only = [a for a in sys.argv[1:] if not a.startswith("-")]
    for c in CASES:
        if only and c[0] not in only:
            continue
        ok &= run_case(*c)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
