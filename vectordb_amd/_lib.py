"""ctypes binding of libepsilla_gfx950.so (C ABI: include/epsilla_gfx950.h).  The library is the only compute
path of this package: if it is missing, or no gfx950 device is usable, we fail loudly — there is no CPU
fallback and nothing under oracle/ is ever imported from here."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libepsilla_gfx950.so")

EPS_OK = 0
_lib = None
_proxy = None
EPS_USER_ERROR = 30000
EPS_INFRA_UNEXPECTED_ERROR = 40001
EPS_DB_UNEXPECTED_ERROR = 50001
EPS_DB_UNSUPPORTED_ERROR = 50002
EPS_NOT_IMPLEMENTED_ERROR = 50009

METRIC_EUCLIDEAN, METRIC_COSINE, METRIC_DOT_PRODUCT = 0, 1, 2
MODE_REFERENCE, MODE_FLAT, MODE_GRAPH = 0, 1, 2
FLAT_AUTO, FLAT_STREAM, FLAT_MFMA, FLAT_MFMA_I8 = 0, 1, 2, 3
FILTER_ROWS_APPEND_ONLY = 1
OPS = {None: 0, "": 0, "<": 1, "<=": 2, "==": 3, "=": 3, ">=": 4, ">": 5, "!=": 6, "<>": 6}

EXPORTS = [
    "eps_default_search_params", "eps_default_build_params", "eps_index_create", "eps_index_create_sharded", "eps_index_destroy",
    "eps_index_last_error", "eps_index_last_error_class", "eps_index_set_stream", "eps_index_synchronize", "eps_index_attach_rows",
    "eps_index_append_rows", "eps_index_attach_shard_rows", "eps_index_clone_rows", "eps_index_row_count", "eps_index_load_table", "eps_index_set_id_map", "eps_index_set_deleted",
    "eps_index_set_int_filter", "eps_index_set_filter_program", "eps_index_set_filter_program_ex", "eps_index_search_walk", "eps_index_select_edges", "eps_index_inter_insert", "eps_index_knn_graph", "eps_index_link", "eps_index_build", "eps_index_set_graph", "eps_index_graph_info",
    "eps_index_get_graph", "eps_index_save_graph", "eps_index_load_graph", "eps_index_search",
    "eps_index_last_stats", "eps_index_kernel_times", "eps_normalize_rows", "eps_merge_topk", "eps_merge_topk_packed", "eps_set_tuning",
    "eps_exchange_unique_id", "eps_exchange_create", "eps_exchange_allgather_merge", "eps_exchange_times", "eps_exchange_info", "eps_exchange_last_error",
    "eps_exchange_create_direct", "eps_exchange_mailbox_export", "eps_exchange_mailbox_connect", "eps_exchange_direct_merge",
    "eps_exchange_destroy",
]

# Engine-selection switches of the library (eps_set_tuning, include/epsilla_gfx950.h).  The library itself reads no environment
# variable; the test-suite and the lab scripts, which steer A/B runs through the environment, opt in with EPS_TUNING_FROM_ENV=1 and this
# wrapper then forwards the names below from os.environ to the library's table before every call.
TUNING_NAMES = (
    "EPS_DEBUG", "EPS_TRV_PROF", "EPS_TRV_PREFILTER", "EPS_TRV_VISITED", "EPS_TRV_STAMP_START", "EPS_TRV_WAVES", "EPS_TRV_WIDE", "EPS_TRV_PER_CU", "EPS_TRV_LDS_KB", "EPS_FLAT_ONE_PASS",
    "EPS_ONE_PASS_TIMED", "EPS_DEBUG_ONE_PASS_OVERFLOW", "EPS_S8_WG_PER_CU", "EPS_S8_HOST_WORDS", "EPS_S8_MAX_Q", "EPS_S8_MAX_K", "EPS_S8_FILTER_PROGRAMS", "EPS_S8_RERANK", "EPS_HOST_STAGING", "EPS_S8_TWO_LAUNCHES", "EPS_RERANK_SPLIT", "EPS_MFMA_BITS", "EPS_MFMA_MAX_BATCH",
    "EPS_MFMA_PROBE", "EPS_MFMA_SEED", "EPS_MFMA_GROUPSYNC", "EPS_MFMA_SYNC_SHIFT", "EPS_MFMA_STAGES", "EPS_MFMA_KERNEL", "EPS_MFMA_NARROW",
    "EPS_MFMA_TWO_PER_CU", "EPS_MFMA_FOLD", "EPS_MFMA_MANTISSA", "EPS_BUILD_BLOCK", "EPS_BUILD_VISITED", "EPS_BUILD_PREFILTER", "EPS_S8_ABLATE", "EPS_S8_FOLD", "EPS_MIRROR_ROTATE", "EPS_MIRROR_CLIP",
)
_forwarded = {}


def set_tuning(name, value):
    """eps_set_tuning: value None removes the entry; name None empties the table."""
    L = load()
    L.eps_set_tuning(None if name is None else name.encode(), None if value is None else str(value).encode())
    if name is None:
        _forwarded.clear()
    else:
        _forwarded.pop(name.encode(), None)


_TUNING_NAMES_B = tuple(n.encode() for n in TUNING_NAMES)


def sync_tuning():
    """Forwards the switches found in the environment (EPS_TUNING_FROM_ENV=1 only: tests, lab scripts)."""
    if os.environ.get("EPS_TUNING_FROM_ENV") != "1" or _lib is None:
        return
    raw = getattr(os.environ, "_data", None)   # (CPython on posix: the bytes -> bytes dict behind os.environ; a lookup there costs 0.1 us, not 0.4)
    if not isinstance(raw, dict):
        raw = {os.fsencode(k): os.fsencode(v) for k, v in os.environ.items() if k.startswith("EPS_")}
    for name in _TUNING_NAMES_B:
        v = raw.get(name)
        if _forwarded.get(name) != v:
            _lib.eps_set_tuning(name, v)
            if v is None:
                _forwarded.pop(name, None)
            else:
                _forwarded[name] = v


class _Synced:
    """The loaded library; calls that may consult a switch forward the environment first (see sync_tuning)."""
    _SYNC = frozenset(("eps_index_search", "eps_index_search_walk", "eps_index_build", "eps_index_knn_graph", "eps_index_link",
                       "eps_index_attach_rows", "eps_index_append_rows", "eps_index_set_graph", "eps_index_load_graph"))

    def __init__(self, cdll):
        object.__setattr__(self, "_cdll", cdll)
        fwd = os.environ.get("EPS_TUNING_FROM_ENV") == "1"
        for name in EXPORTS:
            f = getattr(cdll, name)
            if fwd and name in self._SYNC:
                def call(*a, _f=f):
                    sync_tuning()
                    return _f(*a)
                object.__setattr__(self, name, call)
            else:
                object.__setattr__(self, name, f)

    def __getattr__(self, name):
        return getattr(self._cdll, name)


class SearchParams(C.Structure):
    _fields_ = [("mode", C.c_int32), ("flat_engine", C.c_int32), ("prefilter", C.c_int32),
                ("intra_threads", C.c_int32), ("master_queue", C.c_int64), ("local_queue", C.c_int64),
                ("sync_interval", C.c_int64), ("filter_in_traversal", C.c_int32), ("reserved", C.c_int32)]


class BuildParams(C.Structure):
    _fields_ = [("search_length", C.c_int64), ("out_degree", C.c_int64), ("candidate_pool_size", C.c_int64),
                ("knng", C.c_int64), ("seed", C.c_uint32), ("reserved", C.c_int32)]


class FilterOp(C.Structure):
    _fields_ = [("op", C.c_int32), ("arg", C.c_int32), ("ival", C.c_int64), ("dval", C.c_double)]


FOP = {"const": 1, "dist": 2, "i8": 3, "i16": 4, "i32": 5, "i64": 6, "f32": 7, "f64": 8, "bool": 9, "+": 10, "-": 11, "*": 12, "/": 13,
       "%": 14, "<": 15, "<=": 16, "=": 17, "<>": 18, ">=": 19, ">": 20, "and": 21, "or": 22, "not": 23, "=b": 24, "<>b": 25}


class TableLayout(C.Structure):
    _fields_ = [("primitive_offset", C.c_int64), ("var_len_attrs", C.c_int32), ("dense_fields", C.c_int32),
                ("dense_dims", C.POINTER(C.c_int64)), ("field", C.c_int32), ("reserved", C.c_int32)]


class SearchStats(C.Structure):
    _fields_ = [("dist_evals", C.c_int64), ("expansions", C.c_int64), ("rerank_rows", C.c_int64),
                ("overflow_queries", C.c_int64), ("kernel_ms", C.c_double), ("main_kernel_ms", C.c_double),
                ("main_kernel_launches", C.c_int64), ("main_kernel_rows", C.c_int64),
                ("main_kernel_queries", C.c_int64), ("main_kernel_bits", C.c_int64),
                ("filter_ms_all", C.c_double), ("filter_rows_all", C.c_int64), ("i8_folded", C.c_int64), ("i8_declined", C.c_int64), ("one_pass", C.c_int64), ("i8_rotated", C.c_int64)]


class EpsillaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("[%d] %s" % (code, msg))
        self.code = code


def load():
    """Loads the shared library (no GPU needed for this step) and declares every prototype."""
    global _lib
    if _lib is not None:
        return _synced()
    try:
        # torch ships its own HIP runtime; when both live in one process it must be the first one loaded,
        # otherwise torch finds "no ROCm-capable device".  torch is plumbing here (device memory / streams).
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run `python -m vectordb_amd.build` (hipcc, gfx950). There is no "
                          "fallback implementation." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int32
    L.eps_default_search_params.argtypes = [C.POINTER(SearchParams)]
    L.eps_default_search_params.restype = None
    L.eps_default_build_params.argtypes = [C.POINTER(BuildParams)]
    L.eps_default_build_params.restype = None
    L.eps_index_create.argtypes = [i64, i32, i32, C.POINTER(vp)]
    L.eps_index_create_sharded.argtypes = [i64, i32, C.POINTER(i32), i32, C.POINTER(vp)]
    L.eps_index_destroy.argtypes = [vp]
    L.eps_index_last_error.argtypes = [vp]
    L.eps_index_last_error.restype = C.c_char_p
    L.eps_index_last_error_class.argtypes = [vp]
    L.eps_index_last_error_class.restype = C.c_int32
    L.eps_index_set_stream.argtypes = [vp, vp]
    L.eps_index_synchronize.argtypes = [vp]
    L.eps_index_attach_rows.argtypes = [vp, vp, i64]
    L.eps_index_append_rows.argtypes = [vp, vp, i64]
    L.eps_index_attach_shard_rows.argtypes = [vp, C.c_int32, vp, i64]
    L.eps_index_clone_rows.argtypes = [vp, vp, i64]
    L.eps_index_row_count.argtypes = [vp]
    L.eps_index_row_count.restype = i64
    L.eps_index_load_table.argtypes = [vp, C.c_char_p, C.POINTER(TableLayout), C.POINTER(i64)]
    L.eps_index_set_id_map.argtypes = [vp, i64, i64]
    L.eps_index_set_deleted.argtypes = [vp, vp, i64]
    L.eps_index_set_int_filter.argtypes = [vp, vp, i64, i32, i32, i64]
    L.eps_index_set_filter_program.argtypes = [vp, C.POINTER(FilterOp), i32, vp, i64, i64]
    L.eps_index_set_filter_program_ex.argtypes = [vp, C.POINTER(FilterOp), i32, vp, i64, i64, i32]
    L.eps_index_search_walk.argtypes = [vp, vp, i64, i32, i32, C.POINTER(SearchParams), vp, vp, vp]
    L.eps_index_select_edges.argtypes = [vp, vp, i64, vp, i32, i32, i32, vp, vp]
    L.eps_index_inter_insert.argtypes = [vp, vp, vp, i64, i32, vp, vp]
    L.eps_index_build.argtypes = [vp, i64, C.POINTER(BuildParams)]
    L.eps_index_knn_graph.argtypes = [vp, i64, C.POINTER(BuildParams), vp]
    L.eps_index_link.argtypes = [vp, i64, vp, i64, C.POINTER(BuildParams), vp, vp, C.POINTER(i64)]
    L.eps_index_set_graph.argtypes = [vp, i64, vp, vp, i64]
    L.eps_index_graph_info.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
    L.eps_index_get_graph.argtypes = [vp, vp, vp]
    L.eps_index_save_graph.argtypes = [vp, C.c_char_p]
    L.eps_index_load_graph.argtypes = [vp, C.c_char_p]
    L.eps_index_search.argtypes = [vp, vp, i64, i32, C.POINTER(SearchParams), vp, vp, vp]
    L.eps_index_last_stats.argtypes = [vp, C.POINTER(SearchStats)]
    L.eps_normalize_rows.argtypes = [vp, i64, i64, i32, i32, vp]
    L.eps_merge_topk.argtypes = [vp, vp, i32, i64, i32, vp, vp, i32, vp]
    L.eps_index_kernel_times.argtypes = [vp, C.POINTER(C.c_double), i32]
    L.eps_merge_topk_packed.argtypes = [vp, i64, i64, i32, i64, i32, vp, vp, i32, vp]
    L.eps_set_tuning.argtypes = [C.c_char_p, C.c_char_p]
    L.eps_exchange_unique_id.argtypes = [vp]
    L.eps_exchange_create.argtypes = [i32, i32, vp, i32, C.POINTER(vp)]
    L.eps_exchange_allgather_merge.argtypes = [vp, vp, vp, i64, i32, vp, vp, vp]
    L.eps_exchange_create_direct.argtypes = [i32, i32, i32, C.POINTER(vp)]
    L.eps_exchange_mailbox_export.argtypes = [vp, vp]
    L.eps_exchange_mailbox_connect.argtypes = [vp, vp]
    L.eps_exchange_direct_merge.argtypes = [vp, vp, vp, i64, i32, vp, vp, vp]
    L.eps_exchange_times.argtypes = [vp, C.POINTER(C.c_double), i32]
    L.eps_exchange_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.c_char_p, i64]
    L.eps_exchange_last_error.argtypes = [vp]
    L.eps_exchange_last_error.restype = C.c_char_p
    L.eps_exchange_destroy.argtypes = [vp]
    L.eps_exchange_destroy.restype = None
    for name in EXPORTS:
        if getattr(L, name).restype is C.c_int:
            getattr(L, name).restype = i32
    _lib = L
    return _synced()


def _synced():
    global _proxy
    if _proxy is None:
        _proxy = _Synced(_lib)
    return _proxy
