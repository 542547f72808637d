"""oracle/ref_py.py -- TEST INFRASTRUCTURE ONLY (ctypes view of oracle/_ref/libmsc_ref.so).

The real MeShClust2 reference, compiled from /root/reference by `make -C oracle ref`. Exists only where
that build was possible; `available()` says so. Used to pin the C oracle and to generate tests/golden/.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_ref", "libmsc_ref.so")
NP_T = {8: np.uint8, 16: np.uint16, 32: np.uint32, 64: np.uint64}
_lib = None


def available():
    return os.path.exists(_LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_LIB_PATH)
        vp = C.c_void_p
        L.ref_point_create.restype = vp
        L.ref_point_create.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int]
        L.ref_point_bins.argtypes = [C.c_int, vp, vp]
        L.ref_point_meta.argtypes = [C.c_int, vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.ref_point_clone.restype = vp
        L.ref_point_clone.argtypes = [C.c_int, vp]
        L.ref_point_set.argtypes = [C.c_int, vp, vp]
        L.ref_point_free.argtypes = [C.c_int, vp]
        L.ref_encode.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_uint64)]
        L.ref_raw_feature.restype = C.c_double
        L.ref_raw_feature.argtypes = [C.c_int, C.c_uint64, vp, vp, C.c_int]
        L.ref_distance.restype = C.c_uint64
        L.ref_distance.argtypes = [C.c_int, vp, vp]
        L.ref_model_load.restype = vp
        L.ref_model_load.argtypes = [C.c_int, C.c_char_p]
        L.ref_model_score.argtypes = [C.c_int, vp, vp, vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.ref_model_predict.restype = C.c_double
        L.ref_model_predict.argtypes = [C.c_int, vp, vp, vp]
        L.ref_model_close.argtypes = [C.c_int, vp, vp, vp]
        L.ref_set_bias.argtypes = [C.c_double]
        L.ref_get_close.argtypes = [C.c_int, vp, C.c_double, vp, C.POINTER(vp), C.c_int, C.POINTER(C.c_uint8),
                                    C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.ref_filter.argtypes = [C.c_int, vp, C.c_double, vp, C.POINTER(vp), C.c_int, C.POINTER(C.c_uint8)]
        L.ref_merge.restype = C.c_long
        L.ref_merge.argtypes = [C.c_int, vp, C.c_double, C.POINTER(vp), C.c_int, C.c_long, C.c_long, C.c_long]
        L.ref_mean_nearest.argtypes = [C.c_int, C.POINTER(vp), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        L.ref_set_threads.argtypes = [C.c_int]
        L.ref_train_class.restype = C.c_long
        L.ref_train_class.argtypes = [C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int,
                                      C.c_double, C.c_char_p, C.c_long, C.POINTER(C.c_double)]
        L.ref_train_regr.restype = C.c_long
        L.ref_train_regr.argtypes = [C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_char_p, C.c_long,
                                     C.POINTER(C.c_double)]
        L.ref_time_pairs.restype = C.c_double
        L.ref_time_pairs.argtypes = [C.c_int, vp, C.POINTER(vp), C.c_int, C.c_int, C.POINTER(C.c_double)]
        _lib = L
    return _lib


def _b(s):
    return s if isinstance(s, bytes) else s.encode()


class Point:
    def __init__(self, dtype, seq, k, header=">s", strip=False, handle=None):
        self.dtype = dtype
        self.k = k
        self.h = handle if handle is not None else lib().ref_point_create(dtype, _b(header), _b(seq), k, 1 if strip else 0)
        if not self.h:
            raise ValueError("reference rejected the sequence")

    def bins(self):
        out = np.zeros(4 ** self.k, dtype=NP_T[self.dtype])
        lib().ref_point_bins(self.dtype, self.h, out.ctypes.data)
        return out

    def meta(self):
        mag, ln, sd = C.c_uint64(), C.c_uint64(), C.c_double()
        om = (C.c_uint64 * 4)()
        lib().ref_point_meta(self.dtype, self.h, C.byref(mag), C.byref(ln), C.byref(sd), om)
        return dict(mag=mag.value, length=ln.value, stddev=sd.value, one_mers=list(om))

    def clone(self):
        return Point(self.dtype, None, self.k, handle=lib().ref_point_clone(self.dtype, self.h))

    def set(self, other):
        lib().ref_point_set(self.dtype, self.h, other.h)


def encode(seq):
    seq = _b(seq)
    out = C.create_string_buffer(len(seq) + 1)
    max_segs = len(seq) // 2 + 2
    segs = (C.c_int * (2 * max_segs))()
    eff = C.c_uint64()
    n = lib().ref_encode(b">s", seq, out, segs, max_segs, C.byref(eff))
    if n < 0:
        raise ValueError("invalid nucleotide")
    return out.raw[:len(seq)], [(segs[2 * i], segs[2 * i + 1]) for i in range(n)], eff.value


def raw_feature(flag, a, b):
    return lib().ref_raw_feature(a.dtype, flag, a.h, b.h, a.k)


class Model:
    def __init__(self, dtype, path):
        self.dtype = dtype
        self.h = lib().ref_model_load(dtype, _b(path))
        if not self.h:
            raise ValueError("reference could not load " + path)

    def score(self, a, b):
        singles = (C.c_double * 64)()
        combos = (C.c_double * 64)()
        s, cs = C.c_double(), C.c_double()
        n = lib().ref_model_score(self.dtype, self.h, a.h, b.h, singles, combos, C.byref(s), C.byref(cs))
        if n < 0:
            raise ValueError("reference threw")
        return np.array(singles[:n]), combos, s.value, cs.value

    def predict(self, a, b):
        return lib().ref_model_predict(self.dtype, self.h, a.h, b.h)

    def close(self, a, b):
        return bool(lib().ref_model_close(self.dtype, self.h, a.h, b.h))

    def _parr(self, pts):
        return (C.c_void_p * len(pts))(*[p.h for p in pts])

    def get_close(self, cutoff, query, cands):
        m = len(cands)
        flags = (C.c_uint8 * max(m, 1))()
        bp, bs, im = C.c_int64(), C.c_double(), C.c_int()
        r = lib().ref_get_close(self.dtype, self.h, cutoff, query.h, self._parr(cands), m, flags, C.byref(bp), C.byref(bs), C.byref(im))
        if r != 0:
            raise ValueError("reference threw")
        return np.array(flags[:m], dtype=np.uint8), bp.value, bs.value, bool(im.value)

    def filter(self, cutoff, centre, pts):
        m = len(pts)
        keep = (C.c_uint8 * max(m, 1))()
        lib().ref_filter(self.dtype, self.h, cutoff, centre.h, self._parr(pts), m, keep)
        return np.array(keep[:m], dtype=np.uint8)

    def merge(self, cutoff, centres, current, begin, last):
        return lib().ref_merge(self.dtype, self.h, cutoff, self._parr(centres), len(centres), current, begin, last)


def mean_nearest(pts):
    m = len(pts)
    n = 4 ** pts[0].k
    mean = np.zeros(n, dtype=np.float64)
    d = np.zeros(m, dtype=np.float64)
    nearest = C.c_int64()
    arr = (C.c_void_p * m)(*[p.h for p in pts])
    lib().ref_mean_nearest(pts[0].dtype, arr, m, mean.ctypes.data_as(C.POINTER(C.c_double)),
                           d.ctypes.data_as(C.POINTER(C.c_double)), C.byref(nearest))
    return mean, d, nearest.value


def train_class(dtype, k, first, second, vals, n_train, feat_flags, min_feat, max_feat, ident):
    """BestFirstSelector::train_class on labelled pairs (first[i], second[i], vals[i]); the first n_train are the training set.
    -> (class block of the weights file as text, training accuracy, testing accuracy)"""
    n = len(first)
    vp = C.c_void_p
    fa = (vp * n)(*[p.h for p in first])
    sa = (vp * n)(*[p.h for p in second])
    va = (C.c_double * n)(*[float(v) for v in vals])
    buf = C.create_string_buffer(1 << 16)
    acc = (C.c_double * 2)()
    r = lib().ref_train_class(dtype, fa, sa, va, n_train, n - n_train, k, feat_flags, min_feat, max_feat, ident, buf, len(buf), acc)
    if r < 0:
        raise RuntimeError("reference training failed (%d)" % r)
    return buf.value.decode(), acc[0], acc[1]


def train_regr(dtype, k, first, second, vals, n_train, feat_flags, max_feat):
    """GreedySelector::train_regression followed on the reference's own objects (oracle/ref_harness.cpp train_regr) on labelled pairs
    -> (regression block of the weights file as text, training mean error, testing mean error)"""
    n = len(first)
    vp = C.c_void_p
    fa = (vp * n)(*[p.h for p in first])
    sa = (vp * n)(*[p.h for p in second])
    va = (C.c_double * n)(*[float(v) for v in vals])
    buf = C.create_string_buffer(1 << 16)
    err = (C.c_double * 2)()
    r = lib().ref_train_regr(dtype, fa, sa, va, n_train, n - n_train, k, feat_flags, max_feat, buf, len(buf), err)
    if r < 0:
        raise RuntimeError("reference regression training failed (%d)" % r)
    return buf.value.decode(), err[0], err[1]
