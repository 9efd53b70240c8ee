"""ctypes binding of oracle/liboracle.so (the CPU restatement in oracle/peng_oracle.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "peng_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        u8p, i64p, u64p, f32p, f64p = (C.POINTER(t) for t in (C.c_uint8, C.c_int64, C.c_uint64, C.c_float, C.c_double))
        L.po_revcomp.restype = C.c_uint64
        L.po_revcomp.argtypes = [C.c_uint64, C.c_int]
        L.po_read_fasta.restype = C.c_int64
        L.po_read_fasta.argtypes = [C.c_char_p, C.POINTER(u8p), C.POINTER(i64p)]
        L.po_free.argtypes = [C.c_void_p]
        L.po_count.argtypes = [u8p, i64p, C.c_int64, C.c_int, C.c_int, u64p, u64p]
        L.po_bg_counts.argtypes = [u8p, i64p, C.c_int64, C.c_int, i64p]
        L.po_bg_V.argtypes = [i64p, C.c_int, f32p, f32p]
        L.po_bg_V64.argtypes = [i64p, C.c_int, f32p, f32p]
        L.po_bgprob.argtypes = [C.c_int, C.c_int, f32p, C.c_int, f32p]
        L.po_stats.argtypes = [C.c_int, u64p, f32p, C.c_uint64, f32p, f32p, f32p]
        L.po_select.restype = C.c_int64
        L.po_select.argtypes = [C.c_int, f32p, u64p, C.c_float, C.c_uint64, C.c_int, C.c_int, u64p, C.c_int64]
        L.po_iupac_expand.restype = C.c_int64
        L.po_iupac_expand.argtypes = [C.c_uint64, C.c_int, C.c_int, u64p, C.c_int64]
        L.po_iupac_aggregate.argtypes = [C.c_uint64, C.c_int, C.c_int, u64p, f32p, f32p, C.c_void_p]
        L.po_iupac_count.restype = C.c_uint64
        L.po_iupac_count.argtypes = [C.c_uint64, C.c_int, C.c_int, u64p]
        L.po_mi_score.restype = C.c_float
        L.po_mi_score.argtypes = [C.c_float, C.c_float, C.c_uint]
        L.po_em.restype = C.c_int
        L.po_em.argtypes = [C.c_int, u64p, f32p, f32p, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, f32p]
        L.po_em_accumulate.argtypes = [C.c_int, u64p, f32p, f32p, C.c_float, f64p]
        L.po_synth.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, u8p]
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class IupacStats(C.Structure):
    _fields_ = [("sites", C.c_uint64), ("bg_p", C.c_float), ("expected", C.c_float), ("zscore", C.c_float),
                ("log_pvalue", C.c_float)]


def revcomp(x, W):
    return int(lib().po_revcomp(int(x), W))


def read_fasta(path):
    """-> (codes uint8[total], offs int64[n+1]) as the reference's SequenceSet parses the file."""
    cp = C.POINTER(C.c_uint8)()
    op = C.POINTER(C.c_int64)()
    n = lib().po_read_fasta(path.encode(), C.byref(cp), C.byref(op))
    if n < 0:
        raise ValueError("FASTA error %d" % n)
    offs = np.ctypeslib.as_array(op, shape=(n + 1,)).copy()
    total = int(offs[-1])
    codes = np.ctypeslib.as_array(cp, shape=(max(total, 1),)).copy()[:total]
    lib().po_free(cp)
    lib().po_free(op)
    return codes, offs


def count(codes, offs, W, both):
    codes = np.ascontiguousarray(codes, np.uint8)
    offs = np.ascontiguousarray(offs, np.int64)
    out = np.zeros(4 ** W, np.uint64)
    lt = C.c_uint64(0)
    lib().po_count(_p(codes, C.c_uint8), _p(offs, C.c_int64), len(offs) - 1, W, int(both), _p(out, C.c_uint64), C.byref(lt))
    return out, int(lt.value)


def bg_counts(codes, offs, K=2):
    codes = np.ascontiguousarray(codes, np.uint8)
    offs = np.ascontiguousarray(offs, np.int64)
    out = np.zeros(sum(4 ** (k + 1) for k in range(K + 1)), np.int64)
    lib().po_bg_counts(_p(codes, C.c_uint8), _p(offs, C.c_int64), len(offs) - 1, K, _p(out, C.c_int64))
    return out


def bg_V(n, K=2, alpha=(1.0, 1.0, 1.0), wide=False):
    """wide=False: the reference's `int` counters (wrap beyond 2^31 bases); wide=True: 64-bit counters, the intended
    semantics the product computes -- identical below 2^31 bases."""
    n = np.ascontiguousarray(n, np.int64)
    a = np.asarray(alpha, np.float32)
    V = np.zeros(len(n), np.float32)
    (lib().po_bg_V64 if wide else lib().po_bg_V)(_p(n, C.c_int64), K, _p(a, C.c_float), _p(V, C.c_float))
    return V


def set_threads(n):
    """Threads of the two pattern-space sweeps (bgprob, stats): the loops the reference runs under OpenMP; 1 by default."""
    lib().po_set_threads(int(n))


def bgprob(W, k, V, both):
    V = np.ascontiguousarray(V, np.float32)
    if len(V) < 84:
        V = np.concatenate([V, np.zeros(84 - len(V), np.float32)])
    out = np.zeros(4 ** W, np.float32)
    lib().po_bgprob(W, k, _p(V, C.c_float), int(both), _p(out, C.c_float))
    return out


def stats(W, counts, bgp, ltot):
    counts = np.ascontiguousarray(counts, np.uint64)
    bgp = np.ascontiguousarray(bgp, np.float32)
    e = np.zeros(4 ** W, np.float32)
    lp = np.zeros(4 ** W, np.float32)
    z = np.zeros(4 ** W, np.float32)
    lib().po_stats(W, _p(counts, C.c_uint64), _p(bgp, C.c_float), int(ltot), _p(e, C.c_float), _p(lp, C.c_float), _p(z, C.c_float))
    return e, lp, z


def select(W, z, counts, z_thr=10.0, count_thr=3, single_stranded=False, filter_neighbors=True):
    z = np.ascontiguousarray(z, np.float32)
    counts = np.ascontiguousarray(counts, np.uint64)
    cap = 4 ** W
    seeds = np.zeros(cap, np.uint64)
    n = lib().po_select(W, _p(z, C.c_float), _p(counts, C.c_uint64), z_thr, count_thr, int(single_stranded),
                        int(filter_neighbors), _p(seeds, C.c_uint64), cap)
    return seeds[:n].copy()


def iupac_expand(iupac, W, both):
    cap = 4 ** W
    out = np.zeros(cap, np.uint64)
    n = lib().po_iupac_expand(int(iupac), W, int(both), _p(out, C.c_uint64), cap)
    return out[:n].copy()


def iupac_aggregate(iupac, W, both, counts, bgp, expected):
    counts = np.ascontiguousarray(counts, np.uint64)
    bgp = np.ascontiguousarray(bgp, np.float32)
    expected = np.ascontiguousarray(expected, np.float32)
    st = IupacStats()
    lib().po_iupac_aggregate(int(iupac), W, int(both), _p(counts, C.c_uint64), _p(bgp, C.c_float), _p(expected, C.c_float), C.byref(st))
    return st


def iupac_count(iupac, W, both, counts):
    counts = np.ascontiguousarray(counts, np.uint64)
    return int(lib().po_iupac_count(int(iupac), W, int(both), _p(counts, C.c_uint64)))


def mi_score(obs, exp, nseq):
    return float(lib().po_mi_score(float(obs), float(exp), int(nseq)))


def em(W, counts, bg, pwm, saturation=1e4, threshold=0.08, max_iter=10, mode=0, final_norm=True):
    """-> (pwm float32[W,4], iterations, last change)."""
    counts = np.ascontiguousarray(counts, np.uint64)
    bg = np.ascontiguousarray(bg, np.float32)
    p = np.ascontiguousarray(pwm, np.float32).copy().reshape(W, 4)
    ch = C.c_float(0)
    it = lib().po_em(W, _p(counts, C.c_uint64), _p(bg, C.c_float), _p(p, C.c_float), saturation, threshold, max_iter,
                     mode, int(final_norm), C.byref(ch))
    return p, it, float(ch.value)


def em_accumulate(W, counts, bg, pwm, saturation=1e4):
    counts = np.ascontiguousarray(counts, np.uint64)
    bg = np.ascontiguousarray(bg, np.float32)
    p = np.ascontiguousarray(pwm, np.float32).reshape(W, 4)
    acc = np.zeros((W, 4), np.float64)
    lib().po_em_accumulate(W, _p(counts, C.c_uint64), _p(bg, C.c_float), _p(p, C.c_float), saturation, _p(acc, C.c_double))
    return acc


def synth(seed, seq0, nseq, L):
    out = np.zeros(nseq * L, np.uint8)
    lib().po_synth(seed, seq0, nseq, L, _p(out, C.c_uint8))
    offs = np.arange(nseq + 1, dtype=np.int64) * L
    return out, offs


IUPAC_LETTERS = "ACGTSWRYMKN"


def iupac_id(s):
    return sum(IUPAC_LETTERS.index(c) * 11 ** i for i, c in enumerate(s))


def iupac_str(x, W):
    return "".join(IUPAC_LETTERS[(x // 11 ** i) % 11] for i in range(W))


def kmer_str(x, W):
    return "".join("ACGT"[(x >> (2 * i)) & 3] for i in range(W))
