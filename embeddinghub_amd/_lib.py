"""ctypes binding of the engine's C ABI (include/ehx.h).

The library is the product: if it is missing or no gfx950 device is usable, calls fail loudly
(EhxError / OSError) — there is no Python or CPU fallback for any vector arithmetic.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EHX_LIB") or os.path.join(HERE, "lib", "libehx.so")  # EHX_LIB: ablation builds

OK, EINVAL, ENOTFOUND, EEXISTS, EIMMUTABLE, ENODEVICE, ENOMEM, ERANGE, EUNSUPPORTED, EINTERNAL = (
    0, -1, -2, -3, -4, -5, -6, -7, -8, -9)
METRIC_L2SQ, METRIC_IP, METRIC_COSINE = 0, 1, 2
MODE_FLAT, MODE_GRAPH = 0, 1
DTYPE_F32, DTYPE_F16 = 0, 1
SCAN_AUTO, SCAN_F32, SCAN_F16 = 0, 1, 2
ENGINE_F32, ENGINE_F16, ENGINE_I8 = 0, 1, 2
MAX_K = 48
SEED_CORPUS, SEED_QUERY = 20250211, 20250212


class EhxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ehx error %d: %s" % (code, msg))
        self.code = code


class Params(C.Structure):
    _fields_ = [("mode", C.c_uint32), ("M", C.c_uint32), ("ef_construction", C.c_uint32),
                ("ef", C.c_uint32), ("seed", C.c_uint64), ("initial_capacity", C.c_uint64),
                ("build_batch", C.c_uint32), ("scan", C.c_uint32), ("shards", C.c_uint32),
                ("search_width", C.c_uint32),
                ("reserved", C.c_uint32 * 4)]


class Stats(C.Structure):
    _fields_ = [("n_rows", C.c_uint64), ("capacity", C.c_uint64), ("n_queries", C.c_uint64),
                ("n_dist", C.c_uint64), ("n_hops", C.c_uint64), ("n_rerank", C.c_uint64),
                ("n_uncertified", C.c_uint64), ("bytes_algorithmic", C.c_uint64),
                ("last_scan_ms", C.c_double), ("last_total_ms", C.c_double),
                ("scan_ms_mean", C.c_double), ("scan_launches", C.c_uint64),
                ("n_filter_queries", C.c_uint64), ("n_filter_fallback", C.c_uint64),
                ("n_exhaustive", C.c_uint64), ("n_i8_queries", C.c_uint64), ("n_i8_fallback", C.c_uint64)]


# every symbol include/ehx.h declares: name -> (restype, argtypes)
_f32p, _u64p, _u32p, _i32p = (C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                              C.POINTER(C.c_int32))
_vp = C.c_void_p
SYMBOLS = {
    "ehx_init": (C.c_int, [C.POINTER(C.c_int), C.c_int]),
    "ehx_shutdown": (C.c_int, []),
    "ehx_abi_version": (C.c_int, []),
    "ehx_last_error": (C.c_char_p, []),
    "ehx_space_create": (C.c_int, [C.c_char_p, C.c_size_t, C.c_uint32, C.c_int, C.c_int,
                                   C.POINTER(Params), C.POINTER(_vp)]),
    "ehx_space_open": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(_vp)]),
    "ehx_space_drop": (C.c_int, [_vp]),
    "ehx_space_freeze": (C.c_int, [_vp]),
    "ehx_space_size": (C.c_int, [_vp, _u64p]),
    "ehx_space_dims": (C.c_int, [_vp, _u32p]),
    "ehx_space_reserve": (C.c_int, [_vp, C.c_uint64]),
    "ehx_space_set_ef": (C.c_int, [_vp, C.c_uint32]),
    "ehx_space_set_search_width": (C.c_int, [_vp, C.c_uint32]),
    "ehx_fill_manifold": (C.c_int, [_vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int]),
    "ehx_gen_manifold_rows_device": (C.c_int, [_vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, _vp]),
    "ehx_space_set_scan": (C.c_int, [_vp, C.c_uint32]),
    "ehx_space_scan_engine": (C.c_int, [_vp, _u32p]),
    "ehx_set": (C.c_int, [_vp, C.c_char_p, C.c_size_t, _f32p]),
    "ehx_set_batch": (C.c_int, [_vp, C.c_size_t, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), _f32p]),
    "ehx_get": (C.c_int, [_vp, C.c_char_p, C.c_size_t, _f32p]),
    "ehx_get_by_id": (C.c_int, [_vp, C.c_uint64, _f32p]),
    "ehx_key_of": (C.c_int, [_vp, C.c_uint64, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ehx_knn": (C.c_int, [_vp, C.c_size_t, _f32p, C.c_uint32, _u64p, _f32p, _u32p]),
    "ehx_knn_keys": (C.c_int, [_vp, C.c_size_t, _f32p, C.c_uint32, _u64p, _f32p, _u32p, C.c_char_p,
                               C.c_size_t, _u64p]),
    "ehx_knn_by_key": (C.c_int, [_vp, C.c_char_p, C.c_size_t, C.c_uint32, _u64p, _f32p, _u32p]),
    "ehx_knn_by_key_keys": (C.c_int, [_vp, C.c_char_p, C.c_size_t, C.c_uint32, _u64p, _f32p, _u32p, C.c_char_p, C.c_size_t,
                                      _u64p]),
    "ehx_knn_device": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_uint32, _vp, _vp, _vp]),
    "ehx_merge_topk_device": (C.c_int, [_vp, C.c_size_t, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ehx_merge_topk_strided_device": (C.c_int, [_vp, C.c_size_t, C.c_uint32, C.c_uint32, _vp, C.c_size_t, _vp,
                                                C.c_size_t, _vp, C.c_size_t, _vp, _vp, _vp]),
    "ehx_fill_synthetic": (C.c_int, [_vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int]),
    "ehx_gen_rows_device": (C.c_int, [_vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, _vp]),
    "ehx_graph_import": (C.c_int, [_vp, C.c_uint64, _u32p, _i32p, C.c_uint64, _u32p, _i32p, _u64p, _u32p,
                                   C.c_uint32, C.c_int32]),
    "ehx_graph_export": (C.c_int, [_vp, _u32p, _i32p, _u32p, _u32p, C.c_uint64, _u64p, _u32p, _i32p]),
    "ehx_stats": (C.c_int, [_vp, C.POINTER(Stats)]),
    "ehx_stats_reset": (C.c_int, [_vp]),
    "ehx_graph_counters": (C.c_int, [_vp, _u64p, C.c_uint32]),
}

_LIB = None


def load():
    """Load libehx.so (raises OSError if it was not built: run `python __graft_entry__.py build`)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise OSError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
        # PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64: when both live in one
        # process they must be ONE runtime instance, and that only works if torch's copy is loaded
        # first (ours then resolves to it by SONAME).  Loading libehx first makes a later
        # `import torch` fail with "No HIP GPUs are available".
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(rc):
    if rc != OK:
        raise EhxError(rc, (load().ehx_last_error() or b"").decode(errors="replace"))
    return rc
