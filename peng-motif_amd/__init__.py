"""peng_motif_amd -- ctypes plumbing over libpengk.so (include/pengk.h), the MI355X hot path of PEnG-motif.

The product is the C-ABI library (HIP kernels for gfx950) plus the C++ host mirror of the reference's
classes under peng-motif_amd/host/.  This module only exposes the C ABI to Python so that tests/ and
bench.py can drive it; device memory is either owned here (pengk_malloc) or borrowed from torch
tensors via data_ptr().  There is no CPU fallback: without the built library, or without a GPU,
the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PENGK_LIB: A/B timing of library variants in one session (tools/ab.sh); the product loads the in-tree build
LIB_PATH = os.environ.get("PENGK_LIB") or os.path.join(_HERE, "libpengk.so")
_lib = None

PENGK_OK = 0
ERR_ARG, ERR_DEVICE, ERR_RANGE, ERR_UNSUPPORTED, ERR_NOMEM = 1, 2, 3, 4, 5
FRONT_PAD_BASES = 64

EXPORTS = [
    "pengk_version", "pengk_last_error", "pengk_error_name", "pengk_create", "pengk_destroy", "pengk_synchronize",
    "pengk_stream", "pengk_set_stream", "pengk_set_option", "pengk_get_info", "pengk_malloc", "pengk_free", "pengk_memcpy_h2d", "pengk_memcpy_d2h",
    "pengk_memset", "pengk_warmup", "pengk_host_alloc", "pengk_host_free", "pengk_timer_create", "pengk_timer_record", "pengk_timer_elapsed_ms", "pengk_timer_destroy",
    "pengk_pack", "pengk_pack_threads", "pengk_pack_append", "pengk_packed_free", "pengk_set_sequences", "pengk_synth_sizes", "pengk_synth_sequences",
    "pengk_count", "pengk_count_bg", "pengk_mirror_counts", "pengk_bg_count", "pengk_bg_model", "pengk_pattern_stats",
    "pengk_seed_candidates", "pengk_iupac_aggregate", "pengk_em", "pengk_em_device", "pengk_test_em_generation", "pengk_sequential_sum_f32", "pengk_motif_similarity", "pengk_selftest_division",
    "pengk_comm_unique_id", "pengk_comm_init", "pengk_comm_init_env", "pengk_comm_info", "pengk_comm_init_abandoned", "pengk_comm_rccl_version", "pengk_comm_destroy",
    "pengk_allreduce_tables", "pengk_comm_check_bin_bound", "pengk_allgather",
    "pengk_comm_host_init_env", "pengk_comm_host_info", "pengk_comm_host_allgather", "pengk_comm_host_allreduce_u64",
    "pengk_comm_host_shutdown",
]


class PengkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (lib().pengk_error_name(code).decode(), msg))
        self.code = code


class PackedStruct(C.Structure):
    _fields_ = [("words", C.POINTER(C.c_uint64)), ("n_words", C.c_uint64), ("items", C.POINTER(C.c_uint64)),
                ("n_items", C.c_uint64), ("n_bases", C.c_uint64), ("n_windows", C.c_uint64), ("max_bin_bound", C.c_uint64),
                ("bg_counts", C.c_int64 * 84), ("n_sequences", C.c_uint64), ("max_len", C.c_uint64), ("W", C.c_int),
                ("item_windows", C.c_int), ("all_whole", C.c_int)]


IUPAC_STATS = np.dtype([("sites", np.uint64), ("bg_p", np.float32), ("expected", np.float32), ("zscore", np.float32),
                        ("log_pvalue", np.float32)], align=True)


def lib():
    """Load libpengk.so (built in-tree by peng-motif_amd/build.py).  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libpengk.so is not built: run `python peng-motif_amd/build.py` (needs hipcc)")
        L = C.CDLL(LIB_PATH)
        vp, u64, i64, f32 = C.c_void_p, C.c_uint64, C.c_int64, C.c_float
        L.pengk_version.restype = C.c_int
        L.pengk_last_error.restype = C.c_char_p
        L.pengk_error_name.restype = C.c_char_p
        L.pengk_error_name.argtypes = [C.c_int]
        L.pengk_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.pengk_destroy.argtypes = [vp]
        L.pengk_synchronize.argtypes = [vp]
        L.pengk_stream.restype = vp
        L.pengk_stream.argtypes = [vp]
        L.pengk_set_stream.argtypes = [vp, vp]
        L.pengk_set_option.argtypes = [vp, C.c_char_p, i64]
        L.pengk_get_info.argtypes = [vp, C.c_char_p, C.POINTER(i64)]
        L.pengk_malloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
        L.pengk_free.argtypes = [vp, vp]
        L.pengk_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
        L.pengk_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
        L.pengk_memset.argtypes = [vp, vp, C.c_int, C.c_size_t]
        L.pengk_warmup.argtypes = [vp]
        L.pengk_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
        L.pengk_host_free.argtypes = [vp, vp]
        L.pengk_timer_create.argtypes = [vp, C.POINTER(vp)]
        L.pengk_timer_record.argtypes = [vp, vp]
        L.pengk_timer_elapsed_ms.argtypes = [vp, vp, vp, C.POINTER(f32)]
        L.pengk_timer_destroy.argtypes = [vp, vp]
        L.pengk_pack.argtypes = [vp, vp, i64, C.c_int, C.c_int, C.POINTER(PackedStruct)]
        L.pengk_pack_threads.argtypes = [vp, vp, i64, C.c_int, C.c_int, C.c_int, C.POINTER(PackedStruct)]
        L.pengk_pack_append.argtypes = [vp, vp, i64, C.c_int, C.c_int, vp, C.POINTER(PackedStruct)]
        L.pengk_packed_free.restype = None
        L.pengk_packed_free.argtypes = [C.POINTER(PackedStruct)]
        L.pengk_set_sequences.argtypes = [vp, vp, u64, vp, u64, C.c_int, C.c_int, u64, C.c_int]
        L.pengk_synth_sizes.argtypes = [u64, C.c_uint32, C.c_int, C.c_int, C.POINTER(u64), C.POINTER(u64)]
        L.pengk_synth_sequences.argtypes = [vp, u64, u64, u64, C.c_uint32, C.c_int, C.c_int, vp, vp]
        L.pengk_count.argtypes = [vp, C.c_int, vp, vp]
        L.pengk_count_bg.argtypes = [vp, C.c_int, vp, vp, vp]
        L.pengk_mirror_counts.argtypes = [vp, C.c_int, vp]
        L.pengk_bg_count.argtypes = [vp, vp]
        L.pengk_bg_model.argtypes = [vp, vp, C.c_int, vp, vp]
        L.pengk_pattern_stats.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp]
        L.pengk_seed_candidates.argtypes = [vp, C.c_int, vp, vp, f32, u64, vp, vp, i64, C.POINTER(i64)]
        L.pengk_iupac_aggregate.argtypes = [vp, C.c_int, C.c_int, vp, i64, vp, vp, vp, vp]
        L.pengk_em.argtypes = [vp, C.c_int, i64, vp, f32, f32, C.c_int, vp, vp, vp, vp]
        L.pengk_em_device.argtypes = [vp, C.c_int, i64, vp, f32, f32, C.c_int, vp, vp, vp, vp]
        L.pengk_test_em_generation.argtypes = [vp, C.c_int]
        L.pengk_sequential_sum_f32.argtypes = [vp, vp, u64, u64, vp]
        L.pengk_selftest_division.argtypes = [vp, u64, C.c_uint32, vp]
        L.pengk_motif_similarity.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, C.c_int, vp]
        L.pengk_comm_unique_id.argtypes = [vp]
        L.pengk_comm_init.argtypes = [vp, vp, C.c_int, C.c_int]
        L.pengk_comm_init_env.argtypes = [vp]
        L.pengk_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.pengk_comm_destroy.argtypes = [vp]
        L.pengk_allreduce_tables.argtypes = [vp, C.c_int, vp, vp, vp]
        L.pengk_comm_check_bin_bound.argtypes = [vp]
        L.pengk_allgather.argtypes = [vp, vp, vp, C.c_size_t]
        L.pengk_comm_host_init_env.argtypes = []
        L.pengk_comm_host_info.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.pengk_comm_host_allgather.argtypes = [vp, vp, C.c_size_t]
        L.pengk_comm_host_allreduce_u64.argtypes = [vp, C.c_size_t]
        L.pengk_comm_host_shutdown.argtypes = []
        _lib = L
    return _lib


def _check(rc):
    if rc != PENGK_OK:
        raise PengkError(rc, lib().pengk_last_error().decode())


class Packed:
    """Host result of pengk_pack: numpy copies of the 2-bit stream and the scan items."""

    def __init__(self, codes, offs, W, item_windows=0):
        codes = np.ascontiguousarray(codes, np.uint8)
        offs = np.ascontiguousarray(offs, np.int64)
        st = PackedStruct()
        _check(lib().pengk_pack(codes.ctypes.data, offs.ctypes.data, len(offs) - 1, W, item_windows, C.byref(st)))
        try:
            self.words = np.ctypeslib.as_array(st.words, shape=(st.n_words,)).copy()
            self.items = np.ctypeslib.as_array(st.items, shape=(max(st.n_items, 1),)).copy()[:st.n_items]
            self.n_bases = int(st.n_bases)
            self.n_windows = int(st.n_windows)
            self.max_bin_bound = int(st.max_bin_bound)
            self.bg_counts = np.array(list(st.bg_counts), np.int64)
            self.n_sequences = int(st.n_sequences)
            self.max_len = int(st.max_len)
            self.W = int(st.W)
            self.item_windows = int(st.item_windows)
            self.all_whole = int(st.all_whole)
        finally:
            lib().pengk_packed_free(C.byref(st))


class DeviceArray:
    """Device buffer owned through pengk_malloc."""

    def __init__(self, ctx, shape, dtype):
        self.ctx = ctx
        self.shape = tuple(np.atleast_1d(shape).tolist())
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        _check(lib().pengk_malloc(ctx.h, self.nbytes, C.byref(p)))
        self.ptr = p.value

    @classmethod
    def from_host(cls, ctx, a):
        a = np.ascontiguousarray(a)
        d = cls(ctx, a.shape, a.dtype)
        if d.nbytes:
            _check(lib().pengk_memcpy_h2d(ctx.h, d.ptr, a.ctypes.data, d.nbytes))
        return d

    def to_host(self):
        out = np.empty(self.shape, self.dtype)
        if self.nbytes:
            _check(lib().pengk_memcpy_d2h(self.ctx.h, out.ctypes.data, self.ptr, self.nbytes))
        return out

    def free(self):
        if self.ptr and self.ctx.h:
            lib().pengk_free(self.ctx.h, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _ptr(x):
    """device pointer of a DeviceArray, a torch tensor or a raw int."""
    if isinstance(x, DeviceArray):
        return x.ptr
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    return int(x)


class Context:
    def __init__(self, device=0):
        h = C.c_void_p()
        self.h = None
        _check(lib().pengk_create(device, C.byref(h)))
        self.h = h.value
        self._keep = []
        self.W = None

    def close(self):
        if self.h:
            for k in self._keep:
                if isinstance(k, DeviceArray):
                    k.free()
            self._keep = []
            lib().pengk_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(lib().pengk_synchronize(self.h))

    def set_option(self, name, value):
        _check(lib().pengk_set_option(self.h, name.encode(), int(value)))

    def test_em_generation(self, generation):
        """Test hook: the serial EM by an earlier generation of the library (0 fold, 1 scan; 2 / 3 the current scheme)."""
        _check(lib().pengk_test_em_generation(self.h, int(generation)))

    def info(self, name):
        v = C.c_int64()
        _check(lib().pengk_get_info(self.h, name.encode(), C.byref(v)))
        return int(v.value)

    def empty(self, shape, dtype):
        return DeviceArray(self, shape, dtype)

    def to_device(self, a):
        return DeviceArray.from_host(self, a)

    # ---- timers -----------------------------------------------------------------------------
    def timer(self):
        t = C.c_void_p()
        _check(lib().pengk_timer_create(self.h, C.byref(t)))
        return t.value

    def record(self, t):
        _check(lib().pengk_timer_record(self.h, t))

    def elapsed_ms(self, a, b):
        ms = C.c_float()
        _check(lib().pengk_timer_elapsed_ms(self.h, a, b, C.byref(ms)))
        return float(ms.value)

    # ---- sequences ----------------------------------------------------------------------------
    def set_sequences(self, words, items, n_words, n_items, W, item_windows, max_bin_bound, all_whole):
        _check(lib().pengk_set_sequences(self.h, _ptr(words), n_words, _ptr(items), n_items, W, item_windows,
                                         max_bin_bound, all_whole))
        self._keep = [words, items]
        self.W = W

    def upload(self, packed):
        words = self.to_device(packed.words)
        items = self.to_device(packed.items if len(packed.items) else np.zeros(1, np.uint64))
        self.set_sequences(words, items, len(packed.words), len(packed.items), packed.W, packed.item_windows,
                           packed.max_bin_bound, packed.all_whole)
        self.set_option("n_windows_hint", packed.n_windows)
        return words, items

    def synth(self, seed, seq0, n_seq, L, W, item_windows=0, words=None, items=None):
        nw, ni = C.c_uint64(), C.c_uint64()
        _check(lib().pengk_synth_sizes(n_seq, L, W, item_windows, C.byref(nw), C.byref(ni)))
        if words is None:
            words = self.empty(nw.value, np.uint64)
        if items is None:
            items = self.empty(max(ni.value, 1), np.uint64)
        _check(lib().pengk_synth_sequences(self.h, seed, seq0, n_seq, L, W, item_windows, _ptr(words), _ptr(items)))
        self._keep = [words, items]
        self.W = W
        return words, items, int(nw.value), int(ni.value)

    # ---- kernels ------------------------------------------------------------------------------
    def count(self, both, counts=None, ltot=None):
        if counts is None:
            counts = self.empty(4 ** self.W, np.uint32)
        if ltot is None:
            ltot = self.empty(1, np.uint64)
        _check(lib().pengk_count(self.h, int(both), _ptr(counts), _ptr(ltot)))
        return counts, ltot

    def count_bg(self, both, counts=None, ltot=None, bg=None):
        counts = counts if counts is not None else self.empty(4 ** self.W, np.uint32)
        ltot = ltot if ltot is not None else self.empty(1, np.uint64)
        bg = bg if bg is not None else self.empty(84, np.uint64)
        _check(lib().pengk_count_bg(self.h, int(both), _ptr(counts), _ptr(ltot), _ptr(bg)))
        return counts, ltot, bg

    def mirror(self, W, counts):
        _check(lib().pengk_mirror_counts(self.h, W, _ptr(counts)))

    def bg_count(self, out=None):
        if out is None:
            out = self.empty(84, np.uint64)
        _check(lib().pengk_bg_count(self.h, _ptr(out)))
        return out

    def bg_model(self, bg_counts, K=2, alpha=(1.0, 1.0, 1.0), out=None):
        if out is None:
            out = self.empty(84, np.float32)
        a = np.asarray(alpha, np.float32)
        _check(lib().pengk_bg_model(self.h, _ptr(bg_counts), K, a.ctypes.data, _ptr(out)))
        return out

    def pattern_stats(self, W, both, k, max_k, V, ltot, counts, bgprob=None, expected=None, logp=None, z=None):
        n = 4 ** W
        bgprob = bgprob if bgprob is not None else self.empty((max_k + 1, n), np.float32)
        expected = expected if expected is not None else self.empty(n, np.float32)
        logp = logp if logp is not None else self.empty(n, np.float32)
        z = z if z is not None else self.empty(n, np.float32)
        _check(lib().pengk_pattern_stats(self.h, W, int(both), k, max_k, _ptr(V), _ptr(ltot), _ptr(counts), _ptr(bgprob),
                                         _ptr(expected), _ptr(logp), _ptr(z)))
        return bgprob, expected, logp, z

    def iupac_aggregate(self, W, both, ids, counts, bgp, expected):
        ids = np.ascontiguousarray(ids, np.uint64)
        out = np.zeros(len(ids), IUPAC_STATS)
        assert IUPAC_STATS.itemsize == 24
        _check(lib().pengk_iupac_aggregate(self.h, W, int(both), ids.ctypes.data, len(ids), _ptr(counts), _ptr(bgp),
                                           _ptr(expected), out.ctypes.data))
        return out

    def em(self, W, pwms, counts, bg, saturation=1e4, threshold=0.08, max_iterations=10):
        p = np.ascontiguousarray(pwms, np.float32).copy().reshape(-1, W, 4)
        n = p.shape[0]
        iters = np.zeros(n, np.int32)
        change = np.zeros(n, np.float32)
        _check(lib().pengk_em(self.h, W, n, p.ctypes.data, saturation, threshold, max_iterations, _ptr(counts), _ptr(bg),
                              iters.ctypes.data, change.ctypes.data))
        return p, iters, change

    def sequential_sum(self, terms):
        """left-to-right float32 sums of the rows of a 2-D array (pengk_sequential_sum_f32)"""
        t = np.ascontiguousarray(terms, np.float32)
        t = t.reshape(1, -1) if t.ndim == 1 else t
        d_t = DeviceArray.from_host(self, t.reshape(-1)) if t.size else None
        d_o = DeviceArray.from_host(self, np.zeros(max(t.shape[0], 1), np.float32))
        _check(lib().pengk_sequential_sum_f32(self.h, _ptr(d_t) if d_t is not None else None, t.shape[0], t.shape[1], _ptr(d_o)))
        return d_o.to_host()[:t.shape[0]]

    def em_device(self, W, n_pwm, d_pwms, counts, bg, d_state, d_change, saturation=1e4, threshold=0.08, max_iterations=10):
        _check(lib().pengk_em_device(self.h, W, n_pwm, _ptr(d_pwms), saturation, threshold, max_iterations, _ptr(counts),
                                     _ptr(bg), _ptr(d_state), _ptr(d_change)))
