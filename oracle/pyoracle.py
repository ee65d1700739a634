"""ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front-ends for
  * `Oracle`  -> oracle/libepsilla_oracle.so  (plain-C restatement, oracle/epsilla_oracle.c)
  * `Ref`     -> oracle/_ref/libepsilla_ref.so (the reference's own sources compiled verbatim,
                 oracle/Makefile `ref`; present only where it was built / shipped)
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

try:  # when torch shares the process its bundled HIP runtime must load before ours (libepsilla_dropin links it)
    import torch  # noqa: F401
except ImportError:
    pass

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libepsilla_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libepsilla_ref.so")
DROPIN_SO = os.path.join(os.path.dirname(HERE), "dropin", "_build", "libepsilla_dropin.so")

i64 = C.c_int64
fptr = C.POINTER(C.c_float)
iptr = C.POINTER(C.c_int64)
u8ptr = C.POINTER(C.c_uint8)


def _f(a):
    return a.ctypes.data_as(fptr)


def _i(a):
    return a.ctypes.data_as(iptr)


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])


def build_ref():
    subprocess.check_call(["make", "-s", "-C", HERE, "ref", "-j8"])


class EoCand(C.Structure):
    _fields_ = [("id", C.c_int64), ("dist", C.c_float), ("checked", C.c_uint8)]


class EoFilter(C.Structure):
    _fields_ = [("deleted", C.c_void_p), ("attr", C.c_void_p), ("stride", C.c_int64),
                ("width", C.c_int32), ("op", C.c_int32), ("value", C.c_int64)]


OPS = {None: 0, "": 0, "<": 1, "<=": 2, "==": 3, "=": 3, ">=": 4, ">": 5, "!=": 6, "<>": 6}


def make_filter(deleted=None, attr=None, stride=0, width=4, op=None, value=0, offset=0):
    """Returns (EoFilter, keepalive)."""
    f = EoFilter()
    keep = []
    if deleted is not None:
        deleted = np.ascontiguousarray(deleted, dtype=np.uint8)
        keep.append(deleted)
        f.deleted = deleted.ctypes.data
    if attr is not None and OPS[op]:
        attr = np.ascontiguousarray(attr)
        keep.append(attr)
        f.attr = attr.ctypes.data + offset
        f.stride = stride or attr.strides[0]
        f.width = width
        f.op = OPS[op]
        f.value = int(value)
    return f, keep


class Oracle:
    def __init__(self, path=ORACLE_SO):
        if not os.path.exists(path):
            build_oracle()
        L = self.L = C.CDLL(path)
        L.eo_fvec_l2sqr.restype = C.c_float
        L.eo_fvec_l2sqr.argtypes = [fptr, fptr, i64]
        L.eo_fvec_ip.restype = C.c_float
        L.eo_fvec_ip.argtypes = [fptr, fptr, i64]
        L.eo_dist.restype = C.c_float
        L.eo_dist.argtypes = [C.c_int, fptr, fptr, i64]
        L.eo_dist_batch.argtypes = [C.c_int, fptr, i64, fptr, i64, fptr]
        L.eo_normalize_query.argtypes = [fptr, i64]
        L.eo_normalize_insert.argtypes = [fptr, i64]
        L.eo_bruteforce.restype = i64
        L.eo_bruteforce.argtypes = [C.c_int, fptr, i64, i64, i64, fptr, C.POINTER(EoFilter), C.POINTER(EoCand)]
        L.eo_prepare_init_ids.argtypes = [i64, iptr, iptr, i64, i64, iptr]
        L.eo_search_impl.restype = i64
        L.eo_search_impl.argtypes = [C.c_int, fptr, i64, i64, iptr, iptr, iptr, fptr, C.c_int, i64, i64, i64,
                                     C.POINTER(EoCand), u8ptr]
        L.eo_set_schedule.restype = C.c_int
        L.eo_set_schedule.argtypes = [C.c_int]
        L.eo_search.restype = i64
        L.eo_search.argtypes = [C.c_int, fptr, i64, i64, i64, iptr, iptr, i64, fptr, i64, C.c_int, i64, i64, i64,
                                C.c_int, C.POINTER(EoFilter), iptr, C.POINTER(C.c_double), iptr]
        L.eo_merge_fixed.restype = i64
        L.eo_merge_fixed.argtypes = [C.POINTER(EoCand), i64, C.POINTER(EoCand), i64]
        L.eo_knn_exact.argtypes = [C.c_int, fptr, i64, i64, i64, iptr]
        L.eo_nsg_link.restype = None
        L.eo_nsg_link.argtypes = [fptr, i64, i64, iptr, i64, i64, i64, i64, i64, C.c_uint, iptr, iptr]
        L.eo_nsg_build.restype = i64
        L.eo_nsg_build.argtypes = [fptr, i64, i64, iptr, i64, i64, i64, i64, C.c_uint]
        L.eo_select_edge.restype = i64
        L.eo_select_edge.argtypes = [fptr, i64, i64, iptr, i64, i64, i64, iptr]
        L.eo_inter_insert.restype = None
        L.eo_inter_insert.argtypes = [fptr, i64, i64, iptr, iptr, i64, iptr, iptr]
        L.eo_nsg_fetch.restype = i64
        L.eo_nsg_fetch.argtypes = [iptr, iptr]
        L.eo_graph_file_write.argtypes = [C.c_char_p, i64, i64, iptr, iptr, i64]
        L.eo_graph_file_read.argtypes = [C.c_char_p, iptr, iptr, iptr, iptr, iptr, iptr]

    # ---- distances
    def l2sqr(self, x, y):
        return float(self.L.eo_fvec_l2sqr(_f(x), _f(y), len(x)))

    def ip(self, x, y):
        return float(self.L.eo_fvec_ip(_f(x), _f(y), len(x)))

    def dist(self, metric, row, q):
        return float(self.L.eo_dist(metric, _f(row), _f(q), len(q)))

    def dist_batch(self, metric, rows, q):
        rows = np.ascontiguousarray(rows, np.float32)
        out = np.empty(rows.shape[0], np.float32)
        self.L.eo_dist_batch(metric, _f(rows), rows.shape[0], _f(q), rows.shape[1], _f(out))
        return out

    def normalize_query(self, v):
        v = np.array(v, np.float32)
        self.L.eo_normalize_query(_f(v), len(v))
        return v

    def normalize_insert(self, v):
        v = np.array(v, np.float32)
        self.L.eo_normalize_insert(_f(v), len(v))
        return v

    # ---- flat
    def bruteforce(self, metric, rows, q, start=0, end=None, flt=None):
        rows = np.ascontiguousarray(rows, np.float32)
        end = rows.shape[0] if end is None else end
        out = (EoCand * max(end - start, 1))()
        fp = C.byref(flt) if flt is not None else None
        m = self.L.eo_bruteforce(metric, _f(rows), rows.shape[1], start, end, _f(q), fp, out)
        ids = np.array([out[i].id for i in range(m)], np.int64)
        ds = np.array([out[i].dist for i in range(m)], np.float32)
        return ids, ds

    def topk_flat(self, metric, rows, q, k, flt=None):
        ids, ds = self.bruteforce(metric, rows, q, flt=flt)
        return ids[:k], ds[:k]

    # ---- graph search
    def prepare_init_ids(self, off, nbr, nav, L):
        n = len(off) - 1
        out = np.empty(L, np.int64)
        self.L.eo_prepare_init_ids(n, _i(off), _i(nbr), nav, L, _i(out))
        return out

    def search_impl(self, metric, rows, off, nbr, init_ids, q, T=1, L=500, Lq=None, I=15, lockstep=False):
        """lockstep=True: the T > 1 interleaving the device kernel implements (see eo_search_impl)."""
        Lq = L if Lq is None else Lq
        self.L.eo_set_schedule(int(lockstep))
        rows = np.ascontiguousarray(rows, np.float32)
        n = len(off) - 1
        set_l = (EoCand * ((T - 1) * Lq + L))()
        visited = np.zeros(n, np.uint8)
        ev = self.L.eo_search_impl(metric, _f(rows), rows.shape[1], n, _i(off), _i(nbr), _i(init_ids), _f(q), T, L, Lq,
                                   I, set_l, visited.ctypes.data_as(u8ptr))
        ms = (T - 1) * Lq
        ids = np.array([set_l[ms + i].id for i in range(L)], np.int64)
        ds = np.array([set_l[ms + i].dist for i in range(L)], np.float32)
        return ids, ds, ev

    def search(self, metric, rows, n_indexed, off, nbr, nav, q, limit, T=1, L=500, Lq=None, I=15, prefilter=False,
               flt=None, n_total=None, lockstep=False):
        Lq = L if Lq is None else Lq
        self.L.eo_set_schedule(int(lockstep))
        rows = np.ascontiguousarray(rows, np.float32)
        n_total = rows.shape[0] if n_total is None else n_total
        cap = max(L, limit, 1)
        ids = np.empty(cap, np.int64)
        ds = np.empty(cap, np.float64)
        ev = i64(0)
        fp = C.byref(flt) if flt is not None else None
        if off is None:
            off = np.zeros(1, np.int64)
            nbr = np.zeros(1, np.int64)
        m = self.L.eo_search(metric, _f(rows), rows.shape[1], n_indexed, n_total, _i(off), _i(nbr), nav, _f(q), limit, T,
                             L, Lq, I, int(prefilter), fp, _i(ids), ds.ctypes.data_as(C.POINTER(C.c_double)), C.byref(ev))
        return ids[:m].copy(), ds[:m].copy(), ev.value

    def merge_fixed(self, q1, q2):
        """q1, q2: lists of (id, dist, checked). Returns (merged q1, insert index)."""
        a = (EoCand * len(q1))(*[EoCand(*t) for t in q1])
        b = (EoCand * len(q2))(*[EoCand(*t) for t in q2])
        r = self.L.eo_merge_fixed(a, len(q1), b, len(q2))
        return [(x.id, x.dist, x.checked) for x in a], r

    # ---- build
    def knn_exact(self, metric, rows, K):
        rows = np.ascontiguousarray(rows, np.float32)
        out = np.empty((rows.shape[0], K), np.int64)
        self.L.eo_knn_exact(metric, _f(rows), rows.shape[0], rows.shape[1], K, _i(out))
        return out

    def nsg_build(self, rows, knn, search_length=45, out_degree=50, cand_pool=300, seed=100):
        rows = np.ascontiguousarray(rows, np.float32)
        knn = np.ascontiguousarray(knn, np.int64)
        n = rows.shape[0]
        e = self.L.eo_nsg_build(_f(rows), n, rows.shape[1], _i(knn), knn.shape[1], search_length, out_degree, cand_pool,
                                seed)
        off = np.empty(n + 1, np.int64)
        nbr = np.empty(max(e, 1), np.int64)
        nav = self.L.eo_nsg_fetch(_i(off), _i(nbr))
        return off, nbr[:e], nav

    def nsg_link(self, rows, knn, nav, search_length=45, out_degree=50, cand_pool=300, seed=100):
        """the Link stage alone (GetNeighbors + SyncPrune per node, before InterInsert) on the kNN graph `knn` [n][K] (-1 padded)
        from navigation node `nav`; returns (ids [n][out_degree] -1 padded, deg [n])"""
        rows = np.ascontiguousarray(rows, np.float32)
        knn = np.ascontiguousarray(knn, np.int64)
        n, d = rows.shape
        ids = np.empty((n, out_degree), np.int64)
        deg = np.empty(n, np.int64)
        self.L.eo_nsg_link(_f(rows), n, d, _i(knn), knn.shape[1], search_length, out_degree, cand_pool, int(nav), seed, _i(ids), _i(deg))
        return ids, deg

    def select_edge(self, rows, node, cands, depth=300, out_degree=50):
        rows = np.ascontiguousarray(rows, np.float32)
        cands = np.ascontiguousarray(cands, np.int64)
        out = np.empty(out_degree, np.int64)
        m = self.L.eo_select_edge(_f(rows), rows.shape[1], node, _i(cands), len(cands), depth, out_degree, _i(out))
        return out[:m].copy()

    def inter_insert(self, rows, ids, deg, out_degree):
        """InterInsert over all nodes in order (nsg.cpp:531-536, 583-653) on edge lists ids[n][out_degree] / deg[n]: (ids, deg) after it"""
        rows = np.ascontiguousarray(rows, np.float32)
        ids = np.ascontiguousarray(ids, np.int64)
        deg = np.ascontiguousarray(deg, np.int64)
        out, od = np.empty_like(ids), np.empty_like(deg)
        self.L.eo_inter_insert(_f(rows), rows.shape[0], rows.shape[1], _i(ids), _i(deg), out_degree, _i(out), _i(od))
        return out, od

    def build_graph(self, metric, rows, K=100, **kw):
        """BuildFromVectorTable (ann_graph_segment.cpp:201-242) with exact kNN in place of NN-Descent."""
        K = min(K, rows.shape[0] - 1)
        return self.nsg_build(rows, self.knn_exact(metric, rows, K), **kw)

    def graph_write(self, path, off, nbr, nav, first_id=0):
        return self.L.eo_graph_file_write(path.encode(), len(off) - 1, first_id, _i(off), _i(nbr), nav)

    def graph_read(self, path):
        n, e, fid, nav = i64(0), i64(0), i64(0), i64(0)
        r = self.L.eo_graph_file_read(path.encode(), C.byref(n), C.byref(e), C.byref(fid), None, None, None)
        if r:
            raise IOError("cannot read %s (%d)" % (path, r))
        off = np.empty(n.value + 1, np.int64)
        nbr = np.empty(max(e.value, 1), np.int64)
        r = self.L.eo_graph_file_read(path.encode(), C.byref(n), C.byref(e), C.byref(fid), _i(off), _i(nbr), C.byref(nav))
        if r:
            raise IOError("cannot read %s (%d)" % (path, r))
        return off, nbr[:e.value], nav.value, fid.value


def ref_available():
    return os.path.exists(REF_SO)


def dropin_available():
    return os.path.exists(DROPIN_SO)


def build_dropin():
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(DROPIN_SO).rsplit("/_build", 1)[0], "-j8"])


class Ref:
    """The reference itself (compiled verbatim). See oracle/ref_driver.cpp."""

    def __init__(self, path=REF_SO):
        """path may also be dropin/_build/libepsilla_dropin.so (the reference's DBMS layers over the gfx950
        executor): that build exports only the DBServer-level entry points, so missing low-level symbols are skipped."""
        real = C.CDLL(path)

        class _Tolerant:
            def __getattr__(self_inner, name):
                try:
                    return getattr(real, name)
                except AttributeError:
                    class _Missing:  # accepts restype/argtypes assignments, fails only when called
                        def __call__(self_m, *a, **k):
                            raise AttributeError("%s is not exported by %s" % (name, path))
                    m = _Missing()
                    object.__setattr__(self_inner, name, m)
                    return m

        L = self.L = _Tolerant()
        vp = C.c_void_p
        L.ref_fvec_L2sqr.restype = C.c_float
        L.ref_fvec_L2sqr.argtypes = [fptr, fptr, i64]
        L.ref_fvec_inner_product.restype = C.c_float
        L.ref_fvec_inner_product.argtypes = [fptr, fptr, i64]
        L.ref_dist.restype = C.c_float
        L.ref_dist.argtypes = [C.c_int, fptr, fptr, i64]
        L.ref_dist_batch.argtypes = [C.c_int, fptr, i64, fptr, i64, fptr]
        L.ref_normalize.argtypes = [fptr, i64]
        L.ref_graph_build.restype = vp
        L.ref_graph_build.argtypes = [fptr, i64, i64, C.c_int, C.c_int]
        L.ref_graph_from_arrays.restype = vp
        L.ref_graph_from_arrays.argtypes = [i64, iptr, iptr, i64]
        L.ref_graph_load.restype = vp
        L.ref_graph_load.argtypes = [C.c_char_p, i64, i64]
        L.ref_graph_save.argtypes = [vp, C.c_char_p, i64, i64]
        for nm in ("ref_graph_n", "ref_graph_edges", "ref_graph_nav"):
            getattr(L, nm).restype = i64
            getattr(L, nm).argtypes = [vp]
        L.ref_graph_copy.argtypes = [vp, iptr, iptr]
        L.ref_graph_free.argtypes = [vp]
        L.ref_knn_graph.argtypes = [fptr, i64, i64, i64, C.c_int, C.c_int, iptr]
        L.ref_nsg_from_knn.restype = vp
        L.ref_nsg_from_knn.argtypes = [fptr, i64, i64, iptr, i64, i64, i64, i64, C.c_int, C.c_uint]
        L.ref_select_edge.restype = i64
        L.ref_select_edge.argtypes = [fptr, i64, i64, i64, iptr, i64, i64, i64, iptr]
        if hasattr(L, "ref_inter_insert"):
            L.ref_inter_insert.restype = None
            L.ref_inter_insert.argtypes = [fptr, i64, i64, iptr, iptr, i64, iptr, iptr]
        L.ref_executor_new.restype = vp
        L.ref_executor_new.argtypes = [vp, fptr, i64, C.c_int, C.c_int, i64, i64, i64, C.c_int]
        L.ref_executor_init_ids.argtypes = [vp, iptr]
        L.ref_executor_search_impl.argtypes = [vp, fptr, i64, iptr, fptr]
        L.ref_executor_search_many.restype = C.c_double
        L.ref_executor_search_many.argtypes = [vp, fptr, i64, i64, iptr, fptr]
        L.ref_alloc_rows.restype = C.c_void_p
        L.ref_alloc_rows.argtypes = [i64, i64, C.c_int]
        L.ref_free_rows.argtypes = [C.c_void_p]
        dptr = C.POINTER(C.c_double)
        L.ref_bruteforce_many.restype = C.c_double
        L.ref_bruteforce_many.argtypes = [C.c_void_p, i64, i64, C.c_int, C.c_int, fptr, i64, i64, iptr, fptr, dptr]
        if hasattr(L, "ref_prefilter_many"):
            L.ref_prefilter_many.restype = C.c_double
            L.ref_prefilter_many.argtypes = [C.c_void_p, i64, i64, C.c_int, C.c_int, C.c_void_p, C.c_char_p, fptr, i64, i64, iptr, fptr, iptr, dptr]
        L.ref_pool_search.restype = C.c_double
        L.ref_pool_search.argtypes = [vp, C.c_void_p, i64, C.c_int, C.c_int, C.c_int, i64, i64, fptr, i64, i64, iptr, fptr, dptr]
        L.ref_dist_calls_reset.restype = C.c_uint64
        L.ref_executor_free.argtypes = [vp]
        L.ref_config.argtypes = [C.c_int] * 5
        L.ref_db_new.restype = vp
        L.ref_db_free.argtypes = [vp]
        L.ref_db_load.argtypes = [vp, C.c_char_p, C.c_char_p, i64, C.c_int]
        L.ref_db_create_table.argtypes = [vp, C.c_char_p, C.c_char_p]
        L.ref_db_insert.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_char_p]
        L.ref_db_delete.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]
        L.ref_db_rebuild.argtypes = [vp]
        L.ref_db_swap_executors.argtypes = [vp]
        L.ref_db_search.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, fptr, i64, i64, C.c_char_p,
                                    C.c_int, C.c_char_p, i64]
        if hasattr(L, "ref_db_search_batch"):   # (drop-in build only)
            L.ref_db_search_batch.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, fptr, i64, i64, i64, C.c_char_p, C.c_int, C.c_char_p, i64]
        L.ref_db_search_mt.restype = C.c_double
        L.ref_db_search_mt.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_char_p, fptr, i64, i64, i64, C.c_int, iptr]
        L.ref_db_search_mt_filter.restype = C.c_double
        L.ref_db_search_mt_filter.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_char_p, fptr, i64, i64, i64, C.c_int, C.c_char_p, iptr]
        L.ref_db_get.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, i64, i64, C.c_char_p, i64]

    def l2sqr(self, x, y):
        return float(self.L.ref_fvec_L2sqr(_f(x), _f(y), len(x)))

    def ip(self, x, y):
        return float(self.L.ref_fvec_inner_product(_f(x), _f(y), len(x)))

    def dist(self, metric, row, q):
        return float(self.L.ref_dist(metric, _f(row), _f(q), len(q)))

    def dist_batch(self, metric, rows, q):
        rows = np.ascontiguousarray(rows, np.float32)
        out = np.empty(rows.shape[0], np.float32)
        self.L.ref_dist_batch(metric, _f(rows), rows.shape[0], _f(q), rows.shape[1], _f(out))
        return out

    def normalize(self, v):
        v = np.array(v, np.float32)
        self.L.ref_normalize(_f(v), len(v))
        return v

    def graph_arrays(self, g):
        n = self.L.ref_graph_n(g)
        e = self.L.ref_graph_edges(g)
        off = np.empty(n + 1, np.int64)
        nbr = np.empty(max(e, 1), np.int64)
        self.L.ref_graph_copy(g, _i(off), _i(nbr))
        return off, nbr[:e], self.L.ref_graph_nav(g)

    def build_graph(self, rows, metric=0, threads=1):
        rows = np.ascontiguousarray(rows, np.float32)
        g = self.L.ref_graph_build(_f(rows), rows.shape[0], rows.shape[1], metric, threads)
        return g

    def graph_from_arrays(self, off, nbr, nav):
        off = np.ascontiguousarray(off, np.int64)
        nbr = np.ascontiguousarray(nbr, np.int64)
        return self.L.ref_graph_from_arrays(len(off) - 1, _i(off), _i(nbr), nav)

    def knn_graph(self, rows, K=100, metric=0, threads=1):
        rows = np.ascontiguousarray(rows, np.float32)
        out = np.empty((rows.shape[0], K), np.int64)
        self.L.ref_knn_graph(_f(rows), rows.shape[0], rows.shape[1], K, metric, threads, _i(out))
        return out

    def nsg_from_knn(self, rows, knn, search_length=45, out_degree=50, cand_pool=300, threads=1, seed=100):
        rows = np.ascontiguousarray(rows, np.float32)
        knn = np.ascontiguousarray(knn, np.int64)
        return self.L.ref_nsg_from_knn(_f(rows), rows.shape[0], rows.shape[1], _i(knn), knn.shape[1], search_length,
                                       out_degree, cand_pool, threads, seed)

    def select_edge(self, rows, node, cands, depth=300, out_degree=50):
        rows = np.ascontiguousarray(rows, np.float32)
        cands = np.ascontiguousarray(cands, np.int64)
        out = np.empty(out_degree, np.int64)
        m = self.L.ref_select_edge(_f(rows), rows.shape[0], rows.shape[1], node, _i(cands), len(cands), depth, out_degree, _i(out))
        return out[:m].copy()

    def inter_insert(self, rows, ids, deg, out_degree):
        rows = np.ascontiguousarray(rows, np.float32)
        ids = np.ascontiguousarray(ids, np.int64)
        deg = np.ascontiguousarray(deg, np.int64)
        out, od = np.empty_like(ids), np.empty_like(deg)
        self.L.ref_inter_insert(_f(rows), rows.shape[0], rows.shape[1], _i(ids), _i(deg), out_degree, _i(out), _i(od))
        return out, od

    def executor(self, g, rows, metric=0, T=1, L=500, Lq=None, I=15, count=False):
        Lq = L if Lq is None else Lq
        return self.L.ref_executor_new(g, _f(rows), rows.shape[1], metric, T, L, Lq, I, int(count))

    def init_ids(self, ex, L):
        out = np.empty(L, np.int64)
        self.L.ref_executor_init_ids(ex, _i(out))
        return out

    def search_impl(self, ex, q, K):
        ids = np.empty(K, np.int64)
        ds = np.empty(K, np.float32)
        q = np.ascontiguousarray(q, np.float32)
        self.L.ref_executor_search_impl(ex, _f(q), K, _i(ids), _f(ds))
        return ids, ds

    def search_many(self, ex, Q, K):
        Q = np.ascontiguousarray(Q, np.float32)
        ids = np.empty((Q.shape[0], K), np.int64)
        ds = np.empty((Q.shape[0], K), np.float32)
        sec = self.L.ref_executor_search_many(ex, _f(Q), Q.shape[0], K, _i(ids), _f(ds))
        return ids, ds, sec

    # ---- CPU baseline legs (bench.py)
    def alloc_rows(self, n, d, threads):
        """page-aligned float[n][d] first-touched by `threads` OpenMP threads (NUMA-local to the later scans);
        returns (numpy view, raw pointer); release with free_rows(pointer)."""
        p = self.L.ref_alloc_rows(n, d, threads)
        if not p:
            raise MemoryError("ref_alloc_rows(%d, %d)" % (n, d))
        arr = np.ctypeslib.as_array(C.cast(p, fptr), shape=(n * d,)).reshape(n, d)
        return arr, p

    def free_rows(self, p):
        self.L.ref_free_rows(p)

    def bruteforce_many(self, rows_ptr, n, d, Q, k, metric=0, threads=1):
        """the reference's BruteForceSearch (:717-768) per query; returns ids, dists, per-query seconds"""
        Q = np.ascontiguousarray(Q, np.float32)
        nq = Q.shape[0]
        ids = np.empty((nq, k), np.int64)
        ds = np.empty((nq, k), np.float32)
        sec = np.empty(nq, np.float64)
        self.L.ref_bruteforce_many(rows_ptr, n, d, metric, threads, _f(Q), nq, k, _i(ids), _f(ds),
                                   sec.ctypes.data_as(C.POINTER(C.c_double)))
        return ids, ds, sec

    def prefilter_many(self, rows_ptr, n, d, id_column, flt, Q, k, metric=0, threads=1):
        """the reference's PreFilterBruteForceSearch (:770-831) per query under the filter text `flt` over one INT4 attribute
        "ID" (id_column: int32[n], C-contiguous, kept alive by the caller); returns ids, dists, visible-row counts, seconds"""
        Q = np.ascontiguousarray(Q, np.float32)
        nq = Q.shape[0]
        ids = np.empty((nq, k), np.int64)
        ds = np.empty((nq, k), np.float32)
        cnt = np.empty(nq, np.int64)
        sec = np.empty(nq, np.float64)
        assert id_column.dtype == np.int32 and id_column.flags.c_contiguous and id_column.shape[0] == n
        r = self.L.ref_prefilter_many(rows_ptr, n, d, metric, threads, id_column.ctypes.data_as(C.c_void_p), flt.encode(), _f(Q), nq, k, _i(ids), _f(ds),
                                      _i(cnt), sec.ctypes.data_as(C.POINTER(C.c_double)))
        if r < 0:
            raise ValueError("the reference could not parse the filter %r" % flt)
        return ids, ds, cnt, sec

    def pool_search(self, g, rows_ptr, d, Q, K, E, T, L, I=15, metric=0):
        """E executors x T OpenMP workers (the reference's ExecutorPool concurrency); returns ids, dists, latencies, wall s"""
        Q = np.ascontiguousarray(Q, np.float32)
        nq = Q.shape[0]
        ids = np.empty((nq, K), np.int64)
        ds = np.empty((nq, K), np.float32)
        lat = np.empty(nq, np.float64)
        wall = self.L.ref_pool_search(g, rows_ptr, d, metric, E, T, L, I, _f(Q), nq, K, _i(ids), _f(ds),
                                      lat.ctypes.data_as(C.POINTER(C.c_double)))
        return ids, ds, lat, wall

    # ---- DBServer level
    class DB:
        def __init__(self, ref, path, name="MyDb", scale=150000, wal=True):
            self.r, self.name = ref, name.encode()
            self.h = ref.L.ref_db_new()
            rc = ref.L.ref_db_load(self.h, self.name, path.encode(), scale, int(wal))
            assert rc == 0, rc

        def create_table(self, schema):
            s = schema if isinstance(schema, str) else json.dumps(schema)
            return self.r.L.ref_db_create_table(self.h, self.name, s.encode())

        def insert(self, table, records):
            s = records if isinstance(records, str) else json.dumps(records)
            return self.r.L.ref_db_insert(self.h, self.name, table.encode(), s.encode())

        def delete(self, table, pks, flt=""):
            return self.r.L.ref_db_delete(self.h, self.name, table.encode(), json.dumps(pks).encode(), flt.encode())

        def rebuild(self):
            return self.r.L.ref_db_rebuild(self.h)

        def search(self, table, field, q, limit, fields=("ID",), flt="", with_distance=True, cap=1 << 24):
            q = np.ascontiguousarray(q, np.float32)
            buf = C.create_string_buffer(cap)
            rc = self.r.L.ref_db_search(self.h, self.name, table.encode(), field.encode(), ",".join(fields).encode(),
                                        _f(q), len(q), limit, flt.encode(), int(with_distance), buf, cap)
            txt = buf.value.decode()
            return rc, (json.loads(txt) if rc == 0 else txt)

        def search_batch(self, table, field, Q, limit, fields=("ID",), flt="", with_distance=True, cap=1 << 26):
            """epsdrop::SearchBatch through the drop-in library (C++ level): Q.shape[0] vectors, one device batch"""
            Q = np.ascontiguousarray(Q, np.float32)
            buf = C.create_string_buffer(cap)
            rc = self.r.L.ref_db_search_batch(self.h, self.name, table.encode(), field.encode(), ",".join(fields).encode(), _f(Q), Q.shape[0], Q.shape[1],
                                              limit, flt.encode(), int(with_distance), buf, cap)
            txt = buf.value.decode()
            return rc, (json.loads(txt) if rc == 0 else txt)

        def search_mt(self, table, field, Q, limit, threads, flt=""):
            """Q.shape[0] single-vector Search calls from `threads` client threads; returns (seconds, best ids)."""
            Q = np.ascontiguousarray(Q, np.float32)
            first = np.empty(Q.shape[0], np.int64)
            sec = self.r.L.ref_db_search_mt_filter(self.h, self.name, table.encode(), field.encode(), _f(Q), Q.shape[0], Q.shape[1],
                                                   limit, threads, flt.encode(), _i(first))
            return sec, first

        def get(self, table, fields=("ID",), pks=None, flt="", skip=0, limit=1000, cap=1 << 24):
            buf = C.create_string_buffer(cap)
            rc = self.r.L.ref_db_get(self.h, self.name, table.encode(), ",".join(fields).encode(),
                                     json.dumps(pks or []).encode(), flt.encode(), skip, limit, buf, cap)
            txt = buf.value.decode()
            return rc, (json.loads(txt) if rc == 0 else txt)

        def close(self):
            if self.h:
                self.r.L.ref_db_free(self.h)
                self.h = None

    def db(self, path, **kw):
        return Ref.DB(self, path, **kw)
