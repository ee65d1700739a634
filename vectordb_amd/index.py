"""Host-side mirror of the reference's operator surface for the ANN path, over the C ABI.

  GpuIndex           batch-native handle (one per vector field / shard)
  ANNGraphSegment    same public members and method names as engine/db/ann_graph_segment.hpp:22-55
  VecSearchExecutor  same constructor argument order and result members as
                     engine/db/execution/vec_search_executor.hpp:61-74, 51-52
  GetDistFunc        metric dispatch of engine/db/index/index.cpp:10-35 (returns the metric code the
                     device kernels take; the arithmetic itself lives in csrc/device_common.hpp)

numpy arrays are host buffers; anything exposing `data_ptr()` (a torch tensor on the GPU) is passed as a
device pointer and used in place.
"""
import ctypes as C
import os

import numpy as np

from . import _lib as lib
from ._lib import (EpsillaError, SearchParams, BuildParams, SearchStats, MODE_REFERENCE, MODE_FLAT, MODE_GRAPH,
                   FLAT_AUTO, FLAT_STREAM, FLAT_MFMA, FLAT_MFMA_I8, METRIC_EUCLIDEAN, METRIC_COSINE, METRIC_DOT_PRODUCT, OPS)

METRICS = {"EUCLIDEAN": 0, "COSINE": 1, "DOT_PRODUCT": 2, "L2": 0, "IP": 2, 0: 0, 1: 1, 2: 2}
BruteforceThreshold = 512  # vec_search_executor.hpp:28


def _ptr(a):
    if a is None:
        return None
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    return C.c_void_p(a.ctypes.data)


def _is_dev(a):
    return hasattr(a, "data_ptr")


def GetDistFunc(field_type="VECTOR_FLOAT", metric_type="EUCLIDEAN"):
    """index.cpp:10-35: unknown metrics fall back to L2Sqr."""
    return METRICS.get(metric_type, 0)


class GpuIndex:
    def __init__(self, dim, metric="EUCLIDEAN", device=0, devices=None):
        """devices = [d0, d1, ...]: a hash-sharded index over those GPUs of this process (eps_index_create_sharded; host buffers,
        or device rows per shard / device queries and results on a device of the group); otherwise one index on `device`."""
        self.L = lib.load()
        self.dim = int(dim)
        self.metric = METRICS[metric]
        self.device = device if devices is None else devices[0]
        h = C.c_void_p()
        if devices is not None:
            arr = (C.c_int32 * len(devices))(*devices)
            rc = self.L.eps_index_create_sharded(self.dim, self.metric, arr, len(devices), C.byref(h))
        else:
            rc = self.L.eps_index_create(self.dim, self.metric, device, C.byref(h))
        if rc != 0:
            raise EpsillaError(rc, "eps_index_create failed (no gfx950 device? there is no CPU fallback)")
        self.h = h
        self._keep = {}

    def _check(self, rc):
        if rc != 0:
            raise EpsillaError(rc, self.L.eps_index_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.eps_index_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- data
    def attach_rows(self, rows):
        if not _is_dev(rows):
            rows = np.ascontiguousarray(rows, np.float32)
        assert rows.shape[1] == self.dim
        self._keep["rows"] = rows
        self._check(self.L.eps_index_attach_rows(self.h, _ptr(rows), rows.shape[0]))

    def attach_shard_rows(self, shard, rows):
        """rows of ONE shard of a sharded index (local row l = global row l * shards + shard): numpy, or a tensor on the shard's own
        device (borrowed).  eps_index_attach_shard_rows."""
        if not _is_dev(rows):
            rows = np.ascontiguousarray(rows, np.float32)
        assert rows.shape[1] == self.dim
        self._keep["rows%d" % shard] = rows
        self._check(self.L.eps_index_attach_shard_rows(self.h, int(shard), _ptr(rows), rows.shape[0]))

    def append_rows(self, rows):
        if not _is_dev(rows):
            rows = np.ascontiguousarray(rows, np.float32)
        self._check(self.L.eps_index_append_rows(self.h, _ptr(rows), rows.shape[0]))

    def load_table(self, path, primitive_offset, var_len_attrs, dense_dims, field):
        """eps_index_load_table: the reference's data_mvp.bin straight into HBM; returns the record count"""
        dims = (C.c_int64 * len(dense_dims))(*dense_dims)
        lay = lib.TableLayout(primitive_offset, var_len_attrs, len(dense_dims), dims, field, 0)
        n = C.c_int64()
        self._check(self.L.eps_index_load_table(self.h, path.encode(), C.byref(lay), C.byref(n)))
        return n.value

    @property
    def row_count(self):
        return self.L.eps_index_row_count(self.h)

    def set_id_map(self, base, stride):
        self._check(self.L.eps_index_set_id_map(self.h, base, stride))

    def set_deleted(self, bits):
        if bits is None:
            self._check(self.L.eps_index_set_deleted(self.h, None, 0))
            return
        if not _is_dev(bits):
            bits = np.ascontiguousarray(bits, np.uint8)
        self._keep["deleted"] = bits
        nbytes = bits.numel() if _is_dev(bits) else bits.size
        self._check(self.L.eps_index_set_deleted(self.h, _ptr(bits), nbytes))

    def set_int_filter(self, column, op, value, stride=None, width=None, offset=0):
        """column: numpy array / device tensor holding the attribute rows; the value of row i is the signed
        `width`-byte integer at byte `offset + i*stride` (TableSegmentMVP::attribute_table_ layout)."""
        if column is None or not OPS[op]:
            self._check(self.L.eps_index_set_int_filter(self.h, None, 0, 0, 0, 0))
            return
        if not _is_dev(column):
            column = np.ascontiguousarray(column)
            stride = stride or column.strides[0]
            width = width or column.dtype.itemsize
        else:
            stride = stride or column.element_size()
            width = width or column.element_size()
        self._keep["fcol"] = column
        base = C.c_void_p(_ptr(column).value + offset)
        self._check(self.L.eps_index_set_int_filter(self.h, base, stride, width, OPS[op], int(value)))

    def set_filter_program(self, program, rows=None, stride=None, append_only=False):
        """program: postfix list of ("const", x) | ("dist",) | ("i8"|"i16"|"i32"|"i64"|"f32"|"f64"|"bool", byte_offset) |
        (operator,) with operators + - * / % < <= = <> >= > and or not =b <>b (see eps_filter_op); rows: the packed
        attribute rows (numpy structured/2-D uint8 array or device tensor), row i at i*stride bytes.  None / [] clears.
        Host rows are uploaded in full on every call (edit the array in place and call again to refresh);
        append_only=True promises that rows handed over earlier from the same array are unchanged, so only the new tail is
        uploaded (EPS_FILTER_ROWS_APPEND_ONLY)."""
        if not program:
            self._check(self.L.eps_index_set_filter_program(self.h, None, 0, None, 0, 0))
            return
        ops = (lib.FilterOp * len(program))()
        for i, ins in enumerate(program):
            ops[i].op = lib.FOP[ins[0]]
            if ins[0] == "const":
                ops[i].dval = float(ins[1])
            elif len(ins) > 1:
                ops[i].arg = int(ins[1])
        if rows is None:   # the attribute rows eps_index_load_table kept on the device
            self._check(self.L.eps_index_set_filter_program(self.h, ops, len(program), None, 0, 0))
            return
        if not _is_dev(rows):
            rows = np.ascontiguousarray(rows)
            stride = stride or rows.strides[0]
        n_rows = rows.shape[0]
        self._keep["prog_rows"] = rows
        self._check(self.L.eps_index_set_filter_program_ex(self.h, ops, len(program), _ptr(rows), stride, n_rows,
                                                           lib.FILTER_ROWS_APPEND_ONLY if append_only else 0))

    def search_walk(self, queries, limit, cap, **kw):
        """eps_index_search_walk: the <= cap candidates the reference's post-filter loop would walk (host queries)"""
        p = self.params(**kw)
        queries = np.ascontiguousarray(queries, np.float32)
        if queries.ndim == 1:
            queries = queries[None, :]
        nq = queries.shape[0]
        ids = np.empty((nq, cap), np.int64)
        dist = np.empty((nq, cap), np.float32)
        counts = np.empty(nq, np.int32)
        self._check(self.L.eps_index_search_walk(self.h, _ptr(queries), nq, limit, cap, C.byref(p), _ptr(ids), _ptr(dist), _ptr(counts)))
        return ids, dist, counts

    def set_stream(self, stream_ptr):
        """stream_ptr: a hipStream_t as an integer (torch: `torch.cuda.current_stream().cuda_stream`).  0 is
        torch's default stream = HIP's legacy null stream and is passed as hipStreamLegacy; None gives the index
        its own non-blocking stream again.  Device buffers handed to the index must be ready on ITS stream:
        share the producer's stream (this call) or synchronise the producer first."""
        if stream_ptr is None:
            self._check(self.L.eps_index_set_stream(self.h, None))
        else:
            self._check(self.L.eps_index_set_stream(self.h, C.c_void_p(stream_ptr if stream_ptr else 1)))

    def use_torch_stream(self):
        import torch
        self.set_stream(torch.cuda.current_stream().cuda_stream)
        return self

    def synchronize(self):
        self._check(self.L.eps_index_synchronize(self.h))

    # ---- graph
    def set_graph(self, off, nbr, nav):
        off = np.ascontiguousarray(off, np.int64)
        nbr = np.ascontiguousarray(nbr, np.int64)
        self._check(self.L.eps_index_set_graph(self.h, len(off) - 1, _ptr(off), _ptr(nbr), int(nav)))

    def graph_info(self):
        n, e, nav = C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self.L.eps_index_graph_info(self.h, C.byref(n), C.byref(e), C.byref(nav)))
        return n.value, e.value, nav.value

    def get_graph(self):
        n, e, nav = self.graph_info()
        off = np.zeros(n + 1, np.int64)
        nbr = np.zeros(max(e, 1), np.int64)
        self._check(self.L.eps_index_get_graph(self.h, _ptr(off), _ptr(nbr)))
        return off, nbr[:e], nav

    def build(self, n=None, **kw):
        bp = BuildParams()
        self.L.eps_default_build_params(C.byref(bp))
        for k_, v in kw.items():
            setattr(bp, k_, v)
        self._check(self.L.eps_index_build(self.h, self.row_count if n is None else n, C.byref(bp)))

    def knn_graph(self, n=None, **kw):
        """the kNN-graph stage of the build alone (eps_index_knn_graph): ids [n][min(knng, n-1)], closest first, -1 padded"""
        bp = BuildParams()
        self.L.eps_default_build_params(C.byref(bp))
        for k_, v in kw.items():
            setattr(bp, k_, v)
        n = self.row_count if n is None else n
        out = np.empty((n, min(int(bp.knng), n - 1)), np.int64)
        self._check(self.L.eps_index_knn_graph(self.h, n, C.byref(bp), _ptr(out)))
        return out

    def link(self, knn=None, nav=-1, n=None, **kw):
        """the Link stage alone (eps_index_link) on the kNN graph `knn` [n][K] (None: the device's own) from navigation node `nav`
        (-1: the closest row to the centroid); returns (ids [n][out_degree] -1 padded, deg [n], nav)"""
        bp = BuildParams()
        self.L.eps_default_build_params(C.byref(bp))
        for k_, v in kw.items():
            setattr(bp, k_, v)
        n = self.row_count if n is None else n
        if knn is not None:
            knn = np.ascontiguousarray(knn, np.int64)
            assert knn.shape == (n, min(int(bp.knng), n - 1)), knn.shape
        out = np.empty((n, int(bp.out_degree)), np.int64)
        deg = np.empty(n, np.int32)
        nav_out = C.c_int64(-1)
        self._check(self.L.eps_index_link(self.h, n, _ptr(knn), int(nav), C.byref(bp), _ptr(out), _ptr(deg), C.byref(nav_out)))
        return out, deg, int(nav_out.value)

    def select_edges(self, nodes, cands, depth=300, out_degree=50):
        """SyncPrune's sort + SelectEdge for given candidate lists (eps_index_select_edges); returns (ids [m][R] -1 padded, deg [m])"""
        nodes = np.ascontiguousarray(nodes, np.int64)
        cands = np.ascontiguousarray(cands, np.int64)
        out = np.empty((len(nodes), out_degree), np.int64)
        deg = np.empty(len(nodes), np.int32)
        self._check(self.L.eps_index_select_edges(self.h, _ptr(nodes), len(nodes), _ptr(cands), cands.shape[1], depth, out_degree, _ptr(out), _ptr(deg)))
        return out, deg

    def inter_insert(self, ids, deg, out_degree):
        """The InterInsert stage on given edge lists ids [n][R] / deg [n] (eps_index_inter_insert); returns (ids [n][R] -1 padded, deg [n])"""
        ids = np.ascontiguousarray(ids, np.int64)
        deg = np.ascontiguousarray(deg, np.int32)
        out = np.empty_like(ids)
        od = np.empty(len(deg), np.int32)
        self._check(self.L.eps_index_inter_insert(self.h, _ptr(ids), _ptr(deg), len(deg), out_degree, _ptr(out), _ptr(od)))
        return out, od

    def save_graph(self, path):
        self._check(self.L.eps_index_save_graph(self.h, path.encode()))

    def load_graph(self, path):
        self._check(self.L.eps_index_load_graph(self.h, path.encode()))

    # ---- search
    def params(self, **kw):
        p = SearchParams()
        self.L.eps_default_search_params(C.byref(p))
        for k_, v in kw.items():
            setattr(p, k_, v)
        return p

    def search(self, queries, k, out=None, **kw):
        """Returns (ids int64[nq,k], dist float32[nq,k], counts int32[nq]); numpy for host queries, or the
        caller-provided device tensors `out=(ids, dist, counts)`."""
        p = self.params(**kw)
        if _is_dev(queries):
            nq = queries.shape[0]
            assert out is not None, "device queries need device output tensors: out=(ids, dist, counts)"
            ids, dist, counts = out
        else:
            queries = np.ascontiguousarray(queries, np.float32)
            if queries.ndim == 1:
                queries = queries[None, :]
            nq = queries.shape[0]
            if out is not None:   # (host queries, caller-provided result buffers - host or device: the C ABI takes either side independently)
                ids, dist, counts = out
            else:
                ids = np.empty((nq, k), np.int64)
                dist = np.empty((nq, k), np.float32)
                counts = np.empty(nq, np.int32)
        assert queries.shape[1] == self.dim
        if out is not None:   # the C ABI writes nq x k ids / distances and nq counts through raw pointers: the buffers must be what it assumes
            _check_out(ids, "ids", (nq, k), "int64")
            _check_out(dist, "dist", (nq, k), "float32")
            _check_out(counts, "counts", (nq,), "int32")
        self._check(self.L.eps_index_search(self.h, _ptr(queries), nq, k, C.byref(p), _ptr(ids), _ptr(dist), _ptr(counts)))
        return ids, dist, counts

    def kernel_times(self, cap=64):
        """main-kernel ms of the most recent search calls (oldest first); synchronises the index's stream"""
        buf = (C.c_double * cap)()
        n = self.L.eps_index_kernel_times(self.h, buf, cap)
        return [buf[i] for i in range(n)]

    def stats(self):
        s = SearchStats()
        self._check(self.L.eps_index_last_stats(self.h, C.byref(s)))
        return {f: getattr(s, f) for f, _ in SearchStats._fields_}


def _check_out(a, name, shape, dtype):
    """a caller-provided result buffer (NumPy array or device tensor): dtype, shape and C-contiguity, or a ValueError instead of an out-of-bounds write"""
    got_dtype = str(a.dtype).replace("torch.", "")
    contiguous = a.is_contiguous() if hasattr(a, "is_contiguous") else bool(a.flags["C_CONTIGUOUS"])
    if got_dtype != dtype or tuple(a.shape) != tuple(shape) or not contiguous:
        raise ValueError("search: out[%s] must be a C-contiguous %s array of shape %s, got %s %s%s"
                         % (name, dtype, tuple(shape), got_dtype, tuple(a.shape), "" if contiguous else " (not contiguous)"))


def traversal_gather_bytes(stats, dim, avg_degree, seed_evals=0):
    """Algorithmic bytes the traversal kernel gathers for the search `stats` describes: adjacency lists X*(8 + 4*deg); rows
    E*(4d + 4) when every evaluation reads its fp32 row; with the 8-bit prefilter (stats["rerank_rows"] > 0: fp32 rows read in
    step d) the seed evaluations and the survivors read 4d bytes, every neighbour evaluation its mirror row (d rounded up to 16
    bytes) + the 4-byte row constant.  seed_evals: SearchQueueSize x queries (seeds are always evaluated in fp32)."""
    e, x, f = float(stats["dist_evals"]), float(stats["expansions"]), float(stats["rerank_rows"])
    adj = x * (8 + 4.0 * avg_degree)
    if f <= 0:
        return e * (4.0 * dim + 4) + adj
    q8 = (dim + 15) // 16 * 16
    return (seed_evals + f) * 4.0 * dim + (e - seed_evals) * (q8 + 4.0) + e * 4 + adj


def merge_topk_packed(gathered, shard_stride_bytes, dist_offset_bytes, shards, nq, k, out_dist, out_ids, device=0, stream=None):
    """gathered: one device buffer = the all-gather of every rank's packed [ids int64[nq][k] | dist float[nq][k]]."""
    L = lib.load()
    rc = L.eps_merge_topk_packed(_ptr(gathered), shard_stride_bytes, dist_offset_bytes, shards, nq, k, _ptr(out_dist), _ptr(out_ids),
                                 device, C.c_void_p(stream) if stream else None)
    if rc != 0:
        raise EpsillaError(rc, "eps_merge_topk_packed failed")
    return out_dist, out_ids


class Exchange:
    """eps_exchange (include/epsilla_gfx950.h): the sharded path's one exchange step - an RCCL all-gather of every rank's packed top-k lists and
    the k-way merge - on a communicator the library owns.  Bootstrap: rank 0 calls Exchange.unique_id(), the 128 bytes reach every rank by the
    caller's means, every rank constructs Exchange(rank, world, id_bytes, device) (collective)."""

    @staticmethod
    def unique_id():
        L = lib.load()
        buf = C.create_string_buffer(128)
        rc = L.eps_exchange_unique_id(buf)
        if rc != 0:
            raise EpsillaError(rc, "eps_exchange_unique_id failed (RCCL not loadable?)")
        return buf.raw

    def __init__(self, rank, world, id_bytes, device=0):
        self.L = lib.load()
        assert len(id_bytes) == 128
        self.h = C.c_void_p()
        self.device = device
        rc = self.L.eps_exchange_create(rank, world, C.c_char_p(bytes(id_bytes)), device, C.byref(self.h))
        if rc != 0:
            msg = self.L.eps_exchange_last_error(self.h).decode() if self.h else "eps_exchange_create failed"
            if self.h:
                self.L.eps_exchange_destroy(self.h)
                self.h = None
            raise EpsillaError(rc, msg)

    @classmethod
    def direct(cls, rank, world, device=0):
        """an exchange WITHOUT a communicator: the b = 1 form only (mailbox_export / mailbox_connect / direct_merge); ranks may share a device"""
        self = cls.__new__(cls)
        self.L = lib.load()
        self.h = C.c_void_p()
        self.device = device
        self.world = world
        rc = self.L.eps_exchange_create_direct(rank, world, device, C.byref(self.h))
        if rc != 0:
            msg = self.L.eps_exchange_last_error(self.h).decode() if self.h else "eps_exchange_create_direct failed"
            if self.h:
                self.L.eps_exchange_destroy(self.h)
                self.h = None
            raise EpsillaError(rc, msg)
        return self

    def mailbox_export(self):
        """64 bytes: this rank's mailbox for its peers (hipIpc handle); gather all ranks' handles by your own means and pass them to mailbox_connect"""
        buf = C.create_string_buffer(64)
        rc = self.L.eps_exchange_mailbox_export(self.h, buf)
        if rc != 0:
            raise EpsillaError(rc, self.L.eps_exchange_last_error(self.h).decode())
        return buf.raw

    def mailbox_connect(self, handles):
        """handles: every rank's 64 bytes, in rank order"""
        blob = b"".join(bytes(h) for h in handles)
        assert len(blob) == 64 * len(handles)
        rc = self.L.eps_exchange_mailbox_connect(self.h, C.c_char_p(blob))
        if rc != 0:
            raise EpsillaError(rc, self.L.eps_exchange_last_error(self.h).decode())

    def direct_merge(self, ids, dist, out_ids, out_dist, stream=None):
        """the exchange step without a collective (<= 16 KB of lists per rank): peer stores + flag, device-side wait, merge"""
        nq, k = ids.shape
        rc = self.L.eps_exchange_direct_merge(self.h, _ptr(ids), _ptr(dist), nq, k, _ptr(out_ids), _ptr(out_dist), C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise EpsillaError(rc, self.L.eps_exchange_last_error(self.h).decode())
        return out_ids, out_dist

    def allgather_merge(self, ids, dist, out_ids, out_dist, stream=None):
        """ids int64 [nq][k], dist float32 [nq][k] (this rank's lists, device tensors) -> out_ids / out_dist: the merged global top-k, on every rank"""
        nq, k = ids.shape
        rc = self.L.eps_exchange_allgather_merge(self.h, _ptr(ids), _ptr(dist), nq, k, _ptr(out_ids), _ptr(out_dist), C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise EpsillaError(rc, self.L.eps_exchange_last_error(self.h).decode())
        return out_ids, out_dist

    def times_us(self, max_calls=64):
        """[(all-gather us, merge us)] of the last calls, oldest first (waits for them)"""
        buf = (C.c_double * (2 * max_calls))()
        got = self.L.eps_exchange_times(self.h, buf, max_calls)
        if got < 0:
            raise EpsillaError(-1, "eps_exchange_times failed")
        return [(buf[2 * i], buf[2 * i + 1]) for i in range(got)]

    def info(self):
        r, w, v = C.c_int32(), C.c_int32(), C.c_int32()
        path = C.create_string_buffer(512)
        self.L.eps_exchange_info(self.h, C.byref(r), C.byref(w), C.byref(v), path, 512)
        return {"rank": r.value, "world": w.value, "rccl_version": v.value, "rccl_library": path.value.decode()}

    def close(self):
        if self.h:
            self.L.eps_exchange_destroy(self.h)
            self.h = None


def normalize_rows(rows, only_if_nonzero=True, device=0, stream=None):
    """Normalize (db/vector.cpp:60-69) / insert-time normalisation (table_segment_mvp.cpp:574-587), in place."""
    L = lib.load()
    n, d = rows.shape
    rc = L.eps_normalize_rows(_ptr(rows), n, d, int(only_if_nonzero), device, C.c_void_p(stream) if stream else None)
    if rc != 0:
        raise EpsillaError(rc, "eps_normalize_rows failed")
    return rows


def merge_topk(dist, ids, out_dist, out_ids, device=0, stream=None):
    """dist/ids: [shards, nq, k] (all device tensors or all numpy)."""
    L = lib.load()
    shards, nq, k = dist.shape
    rc = L.eps_merge_topk(_ptr(dist), _ptr(ids), shards, nq, k, _ptr(out_dist), _ptr(out_ids), device,
                          C.c_void_p(stream) if stream else None)
    if rc != 0:
        raise EpsillaError(rc, "eps_merge_topk failed")
    return out_dist, out_ids


# ---------------------------------------------------------------------------------------------------------
class ANNGraphSegment:
    """engine/db/ann_graph_segment.hpp:22-55.  Public members keep the reference's names."""

    def __init__(self, db_catalog_path=None, table_id=None, field_id=None, skip_sync_disk=True):
        self.skip_sync_disk_ = skip_sync_disk if db_catalog_path is None else False
        self.first_record_id_ = 0
        self.record_number_ = 0
        self.offset_table_ = np.zeros(1, np.int64)
        self.neighbor_list_ = np.zeros(0, np.int64)
        self.navigation_point_ = 0
        self._index = None
        if db_catalog_path is not None:  # file ctor, ann_graph_segment.cpp:39-98
            path = self._path(db_catalog_path, table_id, field_id)
            if os.path.exists(path):
                with open(path, "rb") as f:
                    hdr = np.fromfile(f, np.int64, 2)
                    self.record_number_, self.first_record_id_ = int(hdr[0]), int(hdr[1])
                    self.offset_table_ = np.fromfile(f, np.int64, self.record_number_ + 1)
                    self.neighbor_list_ = np.fromfile(f, np.int64, int(self.offset_table_[-1]))
                    self.navigation_point_ = int(np.fromfile(f, np.int64, 1)[0])
            else:
                os.makedirs(os.path.dirname(path), exist_ok=True)
                self.SaveANNGraph(db_catalog_path, table_id, field_id)

    @staticmethod
    def _path(db_catalog_path, table_id, field_id):
        return os.path.join(db_catalog_path, str(table_id), "ann_graph_%d.bin" % field_id)

    def BuildFromVectorTable(self, vector_column, n, dim, metricType, device=0, **build_kw):
        """ann_graph_segment.cpp:201-242, on the device: kNN graph -> NSG -> CSR."""
        ix = GpuIndex(dim, metricType, device)
        ix.attach_rows(vector_column[:n] if not hasattr(vector_column, "data_ptr") else vector_column)
        ix.build(n, **build_kw)
        self.offset_table_, self.neighbor_list_, self.navigation_point_ = ix.get_graph()
        self.record_number_ = n
        self._index = ix
        return self

    def SaveANNGraph(self, db_catalog_path, table_id, field_id, force=False):
        """ann_graph_segment.cpp:156-199 (tmp + fsync + rename); returns a status code."""
        if self.skip_sync_disk_ and not force:
            return 0
        path = self._path(db_catalog_path, table_id, field_id)
        tmp = path + ".tmp"
        try:
            with open(tmp, "wb") as f:
                np.array([self.record_number_, self.first_record_id_], np.int64).tofile(f)
                np.ascontiguousarray(self.offset_table_, np.int64).tofile(f)
                np.ascontiguousarray(self.neighbor_list_, np.int64).tofile(f)
                np.array([self.navigation_point_], np.int64).tofile(f)
                f.flush()
                os.fsync(f.fileno())
            os.rename(tmp, path)
        except OSError:
            return lib.EPS_DB_UNEXPECTED_ERROR
        return 0


class VecSearchExecutor:
    """engine/db/execution/vec_search_executor.hpp:30-74.  One query per Search() call like the reference;
    SearchBatch() is the additive batched entry."""

    def __init__(self, dimension, start_search_point, ann_index, offset_table, neighbor_list, vector_column,
                 fstdistfunc, dist_func_param=None, num_threads=4, L_master=500, L_local=500,
                 subsearch_iterations=15, prefilter_enabled=False, device=0):
        self.ann_index_ = ann_index
        self.total_indexed_vector_ = int(ann_index.record_number_)
        self.dimension_ = int(dimension)
        self.start_search_point_ = int(start_search_point)
        self.num_threads_, self.L_master_, self.L_local_ = num_threads, L_master, L_local
        self.subsearch_iterations_, self.prefilter_enabled_ = subsearch_iterations, prefilter_enabled
        self.brute_force_search_ = self.total_indexed_vector_ < BruteforceThreshold
        self.search_result_ = np.zeros(L_master, np.int64)
        self.distance_ = np.zeros(L_master, np.float64)
        self._ix = GpuIndex(dimension, fstdistfunc, device)
        self._rows = vector_column
        self._attached = -1
        self._graph = (np.asarray(offset_table, np.int64), np.asarray(neighbor_list, np.int64))

    def _sync(self, total_vector, deleted, filter_spec):
        if total_vector != self._attached:
            self._ix.attach_rows(self._rows[:total_vector])
            if self.total_indexed_vector_ > 0:
                self._ix.set_graph(self._graph[0], self._graph[1], self.start_search_point_)
            self._attached = total_vector
        self._ix.set_deleted(deleted)
        if filter_spec:
            self._ix.set_int_filter(*filter_spec)
        else:
            self._ix.set_int_filter(None, None, 0)

    def SearchBatch(self, queries, total_vector, limit, deleted=None, filter_spec=None):
        self._sync(total_vector, deleted, filter_spec)
        ids, dist, counts = self._ix.search(queries, limit, mode=MODE_REFERENCE, prefilter=int(self.prefilter_enabled_),
                                            intra_threads=self.num_threads_, master_queue=self.L_master_,
                                            local_queue=self.L_local_, sync_interval=self.subsearch_iterations_)
        return ids, dist, counts

    def Search(self, query_data, total_vector, limit, deleted=None, filter_spec=None):
        """Status Search(query, table_segment, limit, filter_nodes, result_size): returns (status, result_size);
        results land in search_result_ / distance_ (double), vec_search_executor.cpp:833-935."""
        ids, dist, counts = self.SearchBatch(np.asarray(query_data, np.float32)[None, :], total_vector, limit, deleted,
                                             filter_spec)
        n = int(counts[0])
        if n > len(self.search_result_):
            self.search_result_ = np.zeros(n, np.int64)
            self.distance_ = np.zeros(n, np.float64)
        self.search_result_[:n] = ids[0, :n]
        self.distance_[:n] = dist[0, :n].astype(np.float64)
        return 0, n
