"""ctypes binding of libmsc_driver.so (include/meshclust2_driver.h): the mean-shift clustering LOGIC in host C++ with the hot path
behind callbacks. Pure host code -- it loads without a GPU; what scores the pairs is whatever the callbacks call."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmsc_driver.so")

_vp, _u32, _u64, _i64, _int = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int64, C.c_int
_pu8, _pu32, _pu64, _pi64, _pint = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_int64), C.POINTER(C.c_int)

GET_CLOSE = C.CFUNCTYPE(_int, _vp, _u32, _pu32, _u64, _pu8, _pi64, _pint)
CLOSEST = C.CFUNCTYPE(_int, _vp, _pu32, _u64, _pi64)
CENTRE_NEW = C.CFUNCTYPE(_int, _vp, _u32, _pu32)
CENTRE_SET = C.CFUNCTYPE(_int, _vp, _u32, _u32)
FILTER = C.CFUNCTYPE(_int, _vp, _u32, _pu32, _u64, _pu8)
MERGE = C.CFUNCTYPE(_int, _vp, _pu32, _u64, _i64, _i64, _i64, _pi64)
UPDATE_CENTRES = C.CFUNCTYPE(_int, _vp, _pu32, _u64, _pu32, _pu64, _pi64)
CENTRE_SET_BATCH = C.CFUNCTYPE(_int, _vp, _pu32, _pu32, _u64)
MERGE_ALL = C.CFUNCTYPE(_int, _vp, _pu32, _u64, _int, _pi64)
SET_ORDER = C.CFUNCTYPE(_int, _vp, _pu32, _u64)
GET_CLOSE_RANGE = C.CFUNCTYPE(_int, _vp, _u32, _u64, _u64, _pu32, _pu64, _pi64, _pint)
KILL = C.CFUNCTYPE(_int, _vp, _u64)


class Callbacks(C.Structure):
    _fields_ = [("user", _vp), ("get_close", GET_CLOSE), ("closest", CLOSEST), ("centre_new", CENTRE_NEW), ("centre_set", CENTRE_SET),
                ("filter", FILTER), ("merge", MERGE), ("update_centres", UPDATE_CENTRES), ("centre_set_batch", CENTRE_SET_BATCH),
                ("merge_all", MERGE_ALL)]


class WindowCallbacks(C.Structure):
    _fields_ = [("set_order", SET_ORDER), ("get_close_range", GET_CLOSE_RANGE), ("kill", KILL)]


PROTOTYPES = {
    "msc_cluster_run": (_int, [C.POINTER(Callbacks), _u64, C.POINTER(C.c_char_p), _pu64, C.c_double, _int, _int, C.c_char_p, C.c_char_p, _int,
                               C.c_char_p, C.c_size_t]),
    "msc_cluster_run_windows": (_int, [C.POINTER(Callbacks), C.POINTER(WindowCallbacks), _u64, C.POINTER(C.c_char_p), _pu64, C.c_double, _int, _int, C.c_char_p,
                                       C.c_char_p, _int, C.c_char_p, C.c_size_t]),
    "msc_bins_create": (_vp, [_pu64, _u64, _u64]),
    "msc_bins_destroy": (None, [_vp]),
    "msc_bins_count": (_u64, [_vp]),
    "msc_bins_layout": (_u64, [_vp, _pu32, _pu64]),
    "msc_bins_range": (None, [_vp, _u64, _u64, _pu64]),
    "msc_bins_window": (_i64, [_vp, _u64, _u64, _pu32, _u64]),
    "msc_bins_mark": (None, [_vp, _u64, _u64]),
    "msc_bins_take_marked": (_u64, [_vp, _u64, _u64, _pu32]),
    "msc_bins_take_first": (_i64, [_vp]),
    "msc_bins_erase": (None, [_vp, _u64, _u64]),
    "msc_host_inverse": (None, [_u64, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

_lib = None


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: run `make -C meshclust2_amd/host` (or __graft_entry__.build())" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


class ClusterError(RuntimeError):
    pass


def run(backend, headers, lengths, similarity, delta=5, iterations=15, output=None, log=None, batch_update=True):
    """msc_cluster_run over a Python backend object with the methods of msc::ClusterBackend (msc_driver.hpp):
         get_close(q, window: np.uint32[m]) -> (flags np.uint8[m], pos, is_min)
         closest(members) -> pos          centre_new(point) -> centre        centre_set(centre, point)
         filter(centre, points) -> keep np.uint8[m]          merge(centres, current, begin, last) -> best
       optional: update_centres(centres, points, offsets) -> nearest np.int64[n]; centre_set_batch(centres, points); merge_all(centres, delta) -> best[n]
       optional (all three): set_order(order np.uint32[n]); get_close_range(q, first, end) -> (close positions ascending, best position, is_min); kill(pos)
    An exception raised inside a method aborts the run and is re-raised here."""
    import numpy as np
    lib = load_library()
    failure = []

    def arr(ptr, n, dtype):
        return np.ctypeslib.as_array(ptr, shape=(int(n),)).astype(dtype, copy=True) if n else np.zeros(0, dtype=dtype)

    def guard(fn):
        def wrapped(*a):
            try:
                fn(*a)
                return 0
            except BaseException as e:      # noqa: BLE001 -- nothing may propagate through the C frames
                failure.append(e)
                return -100
        return wrapped

    @guard
    def get_close(_u, q, window, m, flags, pos, is_min):
        f, p, im = backend.get_close(int(q), arr(window, m, np.uint32))
        if m:
            np.ctypeslib.as_array(flags, shape=(int(m),))[:] = f
        pos[0] = int(p)
        is_min[0] = 1 if im else 0

    @guard
    def closest(_u, members, m, pos):
        pos[0] = int(backend.closest(arr(members, m, np.uint32)))

    @guard
    def centre_new(_u, point, centre):
        centre[0] = int(backend.centre_new(int(point)))

    @guard
    def centre_set(_u, centre, point):
        backend.centre_set(int(centre), int(point))

    @guard
    def filter_(_u, centre, points, m, keep):
        k = backend.filter(int(centre), arr(points, m, np.uint32))
        if m:
            np.ctypeslib.as_array(keep, shape=(int(m),))[:] = k

    @guard
    def merge(_u, centres, n, current, begin, last, best):
        best[0] = int(backend.merge(arr(centres, n, np.uint32), int(current), int(begin), int(last)))

    @guard
    def update_centres(_u, centres, n, points, offsets, nearest):
        off = arr(offsets, n + 1, np.uint64)
        res = backend.update_centres(arr(centres, n, np.uint32), arr(points, int(off[-1]) if n else 0, np.uint32), off)
        if n:
            np.ctypeslib.as_array(nearest, shape=(int(n),))[:] = res

    @guard
    def centre_set_batch(_u, centres, points, n):
        backend.centre_set_batch(arr(centres, n, np.uint32), arr(points, n, np.uint32))

    @guard
    def merge_all(_u, centres, n, delta_, best):
        res = backend.merge_all(arr(centres, n, np.uint32), int(delta_))
        if n:
            np.ctypeslib.as_array(best, shape=(int(n),))[:] = res

    @guard
    def set_order(_u, order, n_):
        backend.set_order(arr(order, n_, np.uint32))

    @guard
    def get_close_range(_u, q, first, end, close, n_close, best, is_min):
        c, b, im = backend.get_close_range(int(q), int(first), int(end))
        c = np.asarray(c, dtype=np.uint32)
        if c.size:
            np.ctypeslib.as_array(close, shape=(int(end - first),))[:c.size] = c
        n_close[0] = int(c.size)
        best[0] = int(b)
        is_min[0] = 1 if im else 0

    @guard
    def kill(_u, pos):
        backend.kill(int(pos))

    ranged = all(hasattr(backend, m) for m in ("set_order", "get_close_range", "kill"))
    cb = Callbacks(None, GET_CLOSE(get_close), CLOSEST(closest), CENTRE_NEW(centre_new), CENTRE_SET(centre_set), FILTER(filter_), MERGE(merge),
                   UPDATE_CENTRES(update_centres) if hasattr(backend, "update_centres") else UPDATE_CENTRES(),
                   CENTRE_SET_BATCH(centre_set_batch) if hasattr(backend, "centre_set_batch") else CENTRE_SET_BATCH(),
                   MERGE_ALL(merge_all) if hasattr(backend, "merge_all") else MERGE_ALL())
    wcb = WindowCallbacks(SET_ORDER(set_order), GET_CLOSE_RANGE(get_close_range), KILL(kill)) if ranged else None
    n = len(headers)
    hdr = (C.c_char_p * max(n, 1))(*[h if isinstance(h, bytes) else h.encode() for h in headers])
    lens = (C.c_uint64 * max(n, 1))(*[int(x) for x in lengths])
    err = C.create_string_buffer(512)
    if wcb is not None:
        rc = lib.msc_cluster_run_windows(C.byref(cb), C.byref(wcb), n, hdr, lens, float(similarity), int(delta), int(iterations), output.encode() if output else None,
                                         log.encode() if log else None, 1 if batch_update else 0, err, len(err))
    else:
        rc = lib.msc_cluster_run(C.byref(cb), n, hdr, lens, float(similarity), int(delta), int(iterations), output.encode() if output else None,
                                 log.encode() if log else None, 1 if batch_update else 0, err, len(err))
    if failure:
        raise failure[0]
    if rc != 0:
        raise ClusterError("msc_cluster_run failed (%d): %s" % (rc, err.value.decode(errors="replace")))
