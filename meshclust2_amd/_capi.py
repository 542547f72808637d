"""ctypes binding of libmeshclust2_hip.so (include/meshclust2_hip.h).

The library is the product; this module only declares its prototypes. It fails loudly when the
shared object is missing -- there is no Python/CPU fallback for any entry point.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmeshclust2_hip.so")

MSC_OK = 0
ERR_NAMES = {0: "MSC_OK", -1: "MSC_ERR_INVALID_ARG", -2: "MSC_ERR_NO_DEVICE", -3: "MSC_ERR_HIP", -4: "MSC_ERR_OOM",
             -5: "MSC_ERR_INVALID_INPUT", -6: "MSC_ERR_ZERO_LENGTH", -7: "MSC_ERR_NAN", -8: "MSC_ERR_UNSUPPORTED", -9: "MSC_ERR_IO"}

FEAT = {
    "manhattan": 1 << 2, "euclidean": 1 << 3, "normalized_vectors": 1 << 5, "jefferey_divergence": 1 << 7,
    "pearson": 1 << 9, "intersection": 1 << 13, "emd": 1 << 18, "length_difference": 1 << 21,
    "kulczynski2": 1 << 27, "simratio": 1 << 28, "jensen_shannon": 1 << 29,
    "rre_k_r": 1 << 14, "sim_mm": 1 << 16,          # two `extraslow` statistics (MSC_FEAT_GROUPS)
}
FEAT_FAST = sum(FEAT[n] for n in ("euclidean", "manhattan", "intersection", "kulczynski2", "simratio",
                                  "normalized_vectors", "pearson", "emd", "length_difference"))
FEAT_DIV = FEAT["jefferey_divergence"] | FEAT["jensen_shannon"]
FEAT_SLOW = FEAT_FAST | FEAT_DIV
ORDER_CAND_FIRST, ORDER_QUERY_FIRST = 0, 1
COMBO_XY, COMBO_XY2, COMBO_X2Y, COMBO_X2Y2 = 0, 1, 2, 3


class HistInfo(C.Structure):
    _fields_ = [("mag", C.c_uint64), ("length", C.c_uint64), ("sum", C.c_uint64), ("sum_sq", C.c_uint64),
                ("max_count", C.c_uint64), ("one_mers", C.c_uint64 * 4), ("stddev", C.c_double),
                ("overflow", C.c_int32), ("pad_", C.c_int32), ("id", C.c_uint64)]


# name -> (restype, argtypes); every symbol include/meshclust2_hip.h declares
_vp, _u64, _i64, _int, _dbl = C.c_void_p, C.c_uint64, C.c_int64, C.c_int, C.c_double
_pu8, _pu32, _pu64, _pi64, _pdbl = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_int64), C.POINTER(C.c_double)
PROTOTYPES = {
    "msc_abi_version": (_int, []),
    "msc_create": (_int, [_int, C.POINTER(_vp)]),
    "msc_destroy": (None, [_vp]),
    "msc_last_error": (C.c_char_p, [_vp]),
    "msc_device_name": (_int, [_vp, C.c_char_p, C.c_size_t]),
    "msc_synchronize": (_int, [_vp]),
    "msc_last_kernel_ms": (_int, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "msc_set_kernel_timing": (_int, [_vp, _int]),
    "msc_set_mirror_pass": (_int, [_vp, _int]),
    "msc_set_block_pipe": (_int, [_vp, _int]),
    "msc_last_kernel_launches": (_int, [_vp]),
    "msc_last_kernel_info": (_int, [_vp, C.c_char_p, C.c_size_t, C.POINTER(_int)]),
    "msc_encode": (_int, [C.c_char_p, C.c_size_t, _pu8, _pi64, C.c_size_t, C.POINTER(C.c_size_t), _pu64]),
    "msc_hist_set_create": (_int, [_vp, _int, _int, _u64, C.POINTER(_vp)]),
    "msc_hist_set_create_sparse": (_int, [_vp, _int, _int, _u64, _u64, C.POINTER(_vp)]),
    "msc_hist_set_is_sparse": (_int, [_vp]),
    "msc_hist_set_entries": (_u64, [_vp, _u64]),
    "msc_hist_set_clear": (_int, [_vp, _vp]),
    "msc_hist_set_destroy": (None, [_vp]),
    "msc_hist_set_capacity": (_u64, [_vp]),
    "msc_hist_set_k": (_int, [_vp]),
    "msc_hist_set_dtype": (_int, [_vp]),
    "msc_hist_set_bytes": (_u64, [_vp]),
    "msc_hist_build": (_int, [_vp, _vp, _u64, _u64, C.POINTER(C.c_char_p), _pu64, _int]),
    "msc_hist_build_packed": (_int, [_vp, _vp, _u64, _u64, _vp, _u64, _vp, _vp, _vp, _u64, _vp, _vp]),
    "msc_hist_build_packed_dev": (_int, [_vp, _vp, _u64, _u64, _vp, _u64, _vp, _vp, _vp, _u64, _vp, _vp]),
    "msc_hist_download": (_int, [_vp, _vp, _u64, _vp]),
    "msc_hist_upload": (_int, [_vp, _vp, _u64, _vp, _u64, _pu64]),
    "msc_hist_info_get": (_int, [_vp, _vp, _u64, C.POINTER(HistInfo)]),
    "msc_hist_lengths": (_int, [_vp, _vp, C.c_uint64, C.c_uint64, _vp]),
    "msc_hist_set_id": (_int, [_vp, _vp, _u64, _u64]),
    "msc_hist_clone": (_int, [_vp, _vp, _u64, _vp, _u64]),
    "msc_hist_copy": (_int, [_vp, _vp, _u64, _vp, _u64]),
    "msc_hist_assign": (_int, [_vp, _vp, _u64, _vp, _u64]),
    "msc_hist_assign_batch": (_int, [_vp, _vp, _vp, _vp, _vp, C.c_uint64]),
    "msc_hist_copy_batch": (_int, [_vp, _vp, _vp, _vp, _vp, C.c_uint64]),
    "msc_hist_clone_batch": (_int, [_vp, _vp, _vp, _vp, _vp, C.c_uint64]),
    "msc_model_create": (_int, [_vp, _int, _int, C.POINTER(_int), _pu64, _pdbl, _int, _pu64, _pdbl, _pdbl, _dbl, C.POINTER(_vp)]),
    "msc_model_load": (_int, [_vp, C.c_char_p, _int, C.POINTER(_vp)]),
    "msc_model_parse": (_int, [_vp, C.c_char_p, _int, C.POINTER(_vp)]),
    "msc_model_destroy": (None, [_vp]),
    "msc_model_k": (_int, [_vp]),
    "msc_model_n_singles": (_int, [_vp]),
    "msc_model_n_combos": (_int, [_vp]),
    "msc_model_single_flags": (_int, [_vp, _pu64]),
    "msc_model_set_bias": (None, [_vp, _dbl]),
    "msc_pair_features_raw": (_int, [_vp, _vp, _vp, _u64, _vp, _u64, _int, _u64, _vp]),
    "msc_score": (_int, [_vp, _vp, _vp, _vp, _u64, _vp, _u64, _int, _vp, _vp, _vp, _vp]),
    "msc_score_multi": (_int, [_vp, _vp, _vp, _vp, _u64, _vp, _vp, _u64, _int, _vp, _vp, _vp, _u64, _vp]),
    "msc_get_close": (_int, [_vp, _vp, _dbl, _vp, _vp, _u64, _vp, _u64, _vp, _pi64, _pdbl, C.POINTER(_int)]),
    "msc_filter": (_int, [_vp, _vp, _dbl, _vp, _u64, _vp, _vp, _u64, _vp, _pu64]),
    "msc_merge": (_int, [_vp, _vp, _dbl, _vp, _vp, _u64, _i64, _i64, _i64, _pi64]),
    "msc_search": (_int, [_vp, _vp, _vp, _vp, _vp, _u64, _vp, _u64, _vp, _vp]),
    "msc_mean_nearest": (_int, [_vp, _vp, _vp, _u64, _pi64, _vp, _vp]),
    "msc_update_centres": (_int, [_vp, _vp, C.c_double, _vp, _vp, C.c_uint64, _vp, _vp, _vp, _vp, _vp]),
    "msc_merge_all": (_int, [_vp, _vp, C.c_double, _vp, _vp, C.c_uint64, _int, _vp]),
    "msc_merge_some": (_int, [_vp, _vp, C.c_double, _vp, _vp, C.c_uint64, _int, _vp, C.c_uint64, _vp]),
    "msc_train_class": (_int, [_vp, _vp, _vp, _vp, _vp, C.c_uint64, C.c_uint64, C.c_uint64, _int, _int, C.c_double, C.c_char_p, C.c_size_t, _vp, _vp]),
    "msc_train_regr": (_int, [_vp, _vp, _vp, _vp, _vp, _u64, _u64, _u64, _int, _dbl, C.c_char_p, C.c_size_t, _pdbl, _pdbl]),
    "msc_hist_set_device_view": (_int, [_vp, C.POINTER(_vp), _pu64, C.POINTER(_vp), _pu64]),
    "msc_hist_import_done": (_int, [_vp, _vp, _u64, _u64]),
    "msc_hist_packed_bytes": (_u64, [_vp, _u64]),
    "msc_hist_pack": (_int, [_vp, _vp, _vp, _u64, _vp, _vp]),
    "msc_hist_unpack": (_int, [_vp, _vp, _vp, _u64, _vp, _vp]),
    "msc_hist_set_reset": (_int, [_vp, _vp]),
    "msc_colsum_list_bytes": (_u64, [_vp]),
    "msc_colsum_partial": (_int, [_vp, _vp, _vp, _vp, _u64, C.POINTER(_vp), _pu64]),
    "msc_colsum_nearest": (_int, [_vp, _vp, _vp, _vp, _u64, _vp, _u64, _int, _vp, _vp, _vp]),
    "msc_filter_batch": (_int, [_vp, _vp, _dbl, _vp, _vp, _u64, _vp, _vp, _vp, _vp]),
    "msc_stream_handle": (_vp, [_vp]),
    "msc_device_malloc": (_int, [_vp, _u64, C.POINTER(_vp)]),
    "msc_device_free": (_int, [_vp, _vp]),
    "msc_memcpy_to_host": (_int, [_vp, _vp, _vp, _u64]),
    "msc_memcpy_to_device": (_int, [_vp, _vp, _vp, _u64]),
    "msc_memcpy_device": (_int, [_vp, _vp, _vp, _u64]),
    "msc_last_close_counts": (_int, [_vp, _vp, _u64]),
    "msc_host_alloc": (_int, [_vp, _u64, C.POINTER(_vp)]),
    "msc_host_free": (_int, [_vp, _vp]),
    "msc_window_create": (_int, [_vp, _vp, _vp, _u64, C.POINTER(_vp)]),
    "msc_window_destroy": (None, [_vp]),
    "msc_window_alive": (_u64, [_vp, _u64, _u64]),
    "msc_window_kill": (_int, [_vp, _vp, _vp, _u64]),
    "msc_get_close_window": (_int, [_vp, _vp, _dbl, _vp, _u64, _u64, _vp, _u64, C.POINTER(_pu32), _pu64, _pi64, _pdbl, C.POINTER(_int)]),
}

_lib = None


class MscError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("%s (%d): %s" % (ERR_NAMES.get(code, "?"), code, message))
        self.code = code


def load_library():
    """dlopen the product library and declare every prototype. Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
                          "meshclust2_amd has no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
