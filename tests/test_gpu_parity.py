"""Parity tests proper: the HIP path, called through the C ABI (ctypes -> libepsilla_gfx950.so), against the
CPU oracle on the same seeded inputs and against the golden vectors generated from the compiled reference.
Run with `-m gpu` on an MI355X."""
import os

import numpy as np
import pytest

from helpers import assert_topk_match, bitset, data
from oracle.pyoracle import make_filter

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def amd():
    import vectordb_amd
    from vectordb_amd.build import build
    build()
    return vectordb_amd


# ----------------------------------------------------------------------------------------------- flat scan
@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("n,d", [(1, 4), (5, 4), (300, 7), (1000, 33), (5000, 128), (3000, 768), (2000, 960), (700, 2)])
def test_flat_matches_oracle(amd, oracle, metric, n, d):
    X = data(n, d, 100 + n + d)
    Q = data(5, d, 200 + n + d) * 1.5 - 0.25
    if metric == 1:
        X = np.stack([oracle.normalize_insert(x) for x in X])
        Q = np.stack([oracle.normalize_query(q) for q in Q])
    ix = amd.GpuIndex(d, metric)
    ix.attach_rows(X)
    for k in (1, 10, 100):
        ids, dist, cnt = ix.search(Q, k, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
        for qi, q in enumerate(Q):
            rid, rd = oracle.topk_flat(metric, X, q, k)
            m = int(cnt[qi])
            assert m == len(rid)
            assert_topk_match(ids[qi, :m], dist[qi, :m], rid, rd, what="flat m%d n%d d%d k%d q%d" % (metric, n, d, k, qi))
            assert np.all(ids[qi, m:] == -1) and np.all(np.isinf(dist[qi, m:]))
    ix.close()


@pytest.mark.parametrize("nq", [1, 2, 3, 4, 7, 33])
def test_flat_batch_sizes_and_large_k(amd, oracle, nq):
    X, Q = data(4000, 64, 1), data(nq, 64, 2)
    ix = amd.GpuIndex(64, 0)
    ix.attach_rows(X)
    for k in (10, 65, 200, 600, 1024):
        ids, dist, cnt = ix.search(Q, k, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
        for qi in range(nq):
            rid, rd = oracle.topk_flat(0, X, Q[qi], k)
            assert_topk_match(ids[qi], dist[qi], rid, rd, what="nq%d k%d q%d" % (nq, k, qi))
            assert np.all(np.diff(dist[qi]) >= 0)
    ix.close()


@pytest.mark.parametrize("k", [1025, 3000])
def test_flat_more_than_1024_results_per_query(amd, oracle, k):
    """BruteForceSearch has no result cap (it sorts all n candidates, vec_search_executor.cpp:756-767; DBServer passes the
    caller's `limit` through).  Beyond 1024 results the stream scan pages: page p = the best keys ordered after page p-1's
    last.  Every engine request ends there; with deleted rows and a filter; k > n returns the n visible rows."""
    n, d = 5000, 24
    X, Q = data(n, d, 61), data(3, d, 62)
    bits = bitset(n, range(0, n, 7))
    idc = np.arange(n, dtype=np.int64)
    ix = amd.GpuIndex(d, 0)
    ix.attach_rows(X)
    for eng in (amd.FLAT_AUTO, amd.FLAT_STREAM, amd.FLAT_MFMA, amd.FLAT_MFMA_I8):
        ids, dist, cnt = ix.search(Q, k, mode=amd.MODE_FLAT, flat_engine=eng)
        for qi in range(len(Q)):
            rid, rd = oracle.topk_flat(0, X, Q[qi], k)
            assert int(cnt[qi]) == k
            assert_topk_match(ids[qi], dist[qi], rid, rd, what="k%d q%d" % (k, qi))
    ix.set_deleted(bits)
    ix.set_int_filter(idc, "<", 4000)
    flt, keep = make_filter(deleted=bits, attr=idc, stride=8, width=8, op="<", value=4000)
    ids, dist, cnt = ix.search(Q, 4000, mode=amd.MODE_FLAT)
    for qi in range(len(Q)):
        rid, rd = oracle.topk_flat(0, X, Q[qi], 4000, flt=flt)
        assert int(cnt[qi]) == len(rid) < 4000
        assert_topk_match(ids[qi][:len(rid)], dist[qi][:len(rid)], rid, rd, what="filtered q%d" % qi)
        assert (ids[qi][len(rid):] == -1).all()
    ix.close()


def test_flat_deleted_and_filter(amd, oracle):
    n, d = 3000, 48
    X, Q = data(n, d, 3), data(6, d, 4)
    idcol = np.arange(n, dtype=np.int32)
    attr = np.zeros((n, 3), np.int32)  # packed attribute rows: the ID lives at byte offset 4, stride 12
    attr[:, 1] = idcol
    dele = list(range(0, n, 2))
    bits = bitset(n, dele)
    ix = amd.GpuIndex(d, 0)
    ix.attach_rows(X)
    ix.set_deleted(bits)
    for op, val in (("<", 1500), (">=", 2990), ("=", 77), ("!=", 5), ("<=", 2), (">", 10 ** 6)):
        ix.set_int_filter(attr, op, val, stride=12, width=4, offset=4)
        flt, keep = make_filter(deleted=bits, attr=attr, stride=12, width=4, op=op, value=val, offset=4)
        ids, dist, cnt = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
        for qi, q in enumerate(Q):
            rid, rd = oracle.topk_flat(0, X, q, 10, flt=flt)
            m = int(cnt[qi])
            assert m == len(rid)
            assert_topk_match(ids[qi, :m], dist[qi, :m], rid, rd, what="%s %d" % (op, val))
    # prefilter mode gives the same rows (PreFilterBruteForceSearch, :770-831)
    ix.set_int_filter(attr, "<", 1500, stride=12, width=4, offset=4)
    a = ix.search(Q, 10, mode=amd.MODE_REFERENCE, prefilter=1)
    b = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
    assert np.array_equal(a[0], b[0])
    ix.close()


def test_reference_mode_small_table_caps_at_local_queue(amd, oracle):
    """n_indexed < 512 -> BruteForceSearch; result_size = min(survivors, limit, L_local) (:862-868)."""
    X, Q = data(400, 16, 5), data(3, 16, 6)
    ix = amd.GpuIndex(16, 0)
    ix.attach_rows(X)
    ids, dist, cnt = ix.search(Q, 300, mode=amd.MODE_REFERENCE, local_queue=120, master_queue=120)
    assert list(cnt) == [120, 120, 120]
    for qi, q in enumerate(Q):
        oid, od, _ = oracle.search(0, X, 0, None, None, 0, q, 300, L=120)
        assert_topk_match(ids[qi, :120], dist[qi, :120], oid, od)
    ix.close()


def test_known_answers_through_executor_mirror(amd, oracle):
    """The reference's own gtest expectations (db_server.cpp:289-292, :514-751, :1407-1630), driven through the
    VecSearchExecutor mirror class."""
    cities = [("Berlin", [0.05, 0.61, 0.76, 0.74]), ("London", [0.19, 0.81, 0.75, 0.11]),
              ("Moscow", [0.36, 0.55, 0.47, 0.94]), ("San Francisco", [0.18, 0.01, 0.85, 0.80]),
              ("Shanghai", [0.24, 0.18, 0.22, 0.44])]
    X = np.array([v for _, v in cities], np.float32)
    q = np.array([0.35, 0.55, 0.47, 0.94], np.float32)
    want = {"EUCLIDEAN": ["Moscow", "Berlin", "Shanghai", "San Francisco", "London"],
            "DOT_PRODUCT": ["Moscow", "Berlin", "San Francisco", "London", "Shanghai"],
            "COSINE": ["Moscow", "Shanghai", "Berlin", "San Francisco", "London"]}
    seg = amd.ANNGraphSegment()
    for metric, order in want.items():
        rows, qq = X.copy(), q.copy()
        if metric == "COSINE":
            amd.normalize_rows(rows, only_if_nonzero=True)
            qq = amd.normalize_rows(qq[None, :].copy(), only_if_nonzero=False)[0]
        ex = amd.VecSearchExecutor(4, 0, seg, seg.offset_table_, seg.neighbor_list_, rows, amd.GetDistFunc("VECTOR_FLOAT", metric))
        st, n = ex.Search(qq, 5, 6)
        assert st == 0 and n == 5
        assert [cities[i][0] for i in ex.search_result_[:n]] == order
        if metric == "EUCLIDEAN":
            assert abs(ex.distance_[0] - 1.0000040e-4) < 1e-8
            st, n = ex.Search(qq, 5, 6, deleted=np.array([0b1111], np.uint8))
            assert [cities[i][0] for i in ex.search_result_[:n]] == ["Shanghai"]
            assert abs(ex.distance_[0] - 0.46149999) < 1e-6
            st, n = ex.Search(qq, 5, 6, filter_spec=(np.arange(1, 6, dtype=np.int32), "<=", 2))
            assert sorted(int(i) + 1 for i in ex.search_result_[:n]) == [1, 2]
        if metric == "DOT_PRODUCT":
            assert abs(ex.distance_[0] + 1.5329999924) < 1e-6


def test_normalize_matches_reference_semantics(amd, oracle):
    V = data(50, 37, 8) - 0.5
    V[7] = 0.0
    a = amd.normalize_rows(V.copy(), only_if_nonzero=True)
    for i in range(50):
        assert np.allclose(a[i], oracle.normalize_insert(V[i]), rtol=2e-6, atol=1e-7)
    assert np.all(a[7] == 0)
    b = amd.normalize_rows(V[:3].copy(), only_if_nonzero=False)
    for i in range(3):
        assert np.allclose(b[i], oracle.normalize_query(V[i]), rtol=2e-6, atol=1e-7)


# ----------------------------------------------------------------------------------------------- graph search
def _golden_graph():
    z = np.load(os.path.join(G, "graph2000x32.npz"))
    return z, z["off"].astype(np.int64), z["nbr"].astype(np.int64), int(z["nav"])


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_graph_search_T1_matches_reference_golden(amd, oracle, metric):
    """Same graph (built by the reference), IntraQueryThreads = 1: the device traversal must return the
    reference's result; golden = VecSearchExecutor::SearchImpl master queue from the compiled reference."""
    z, off, nbr, nav = _golden_graph()
    X, Q = data(2000, 32, 42), data(16, 32, 43)
    ix = amd.GpuIndex(32, metric)
    ix.attach_rows(X)
    ix.set_graph(off, nbr, nav)
    for k in (10, 100, 500):
        ids, dist, cnt = ix.search(Q, k, mode=amd.MODE_GRAPH, intra_threads=1)
        assert np.all(cnt == k)
        for qi in range(len(Q)):
            assert_topk_match(ids[qi], dist[qi], z["ids_m%d" % metric][qi][:k], z["dist_m%d" % metric][qi][:k],
                              what="graph m%d k%d q%d" % (metric, k, qi))
    # distance evaluations: identical visit sequence => identical count (up to fp ties at the bound)
    ix.search(Q[:1], 10, mode=amd.MODE_GRAPH, intra_threads=1)
    ev_gpu = ix.stats()["dist_evals"]
    init = oracle.prepare_init_ids(off, nbr, nav, 500)
    _, _, ev_or = oracle.search_impl(metric, X, off, nbr, init, Q[0], T=1, L=500)
    assert abs(ev_gpu - ev_or) <= max(2, ev_or // 200), (ev_gpu, ev_or)
    ix.close()


def test_graph_search_T4_reaches_exact_answer(amd, oracle):
    z, off, nbr, nav = _golden_graph()
    X, Q = data(2000, 32, 42), data(16, 32, 44)
    ix = amd.GpuIndex(32, 0)
    ix.attach_rows(X)
    ix.set_graph(off, nbr, nav)
    ids, dist, cnt = ix.search(Q, 10, mode=amd.MODE_GRAPH, intra_threads=4)
    for qi, q in enumerate(Q):
        rid, rd = oracle.topk_flat(0, X, q, 10)
        assert_topk_match(ids[qi], dist[qi], rid, rd, what="T4 q%d" % qi)
    ix.close()


def test_search_graph_tail_filter_deleted_golden(amd, oracle):
    """Full Search(): graph over [0,1000) + brute-force tail [1000,1500) + merge + post-filter + deletes, against
    the results the reference's DBServer returned (tests/golden/dbserver1500x8.npz)."""
    z = np.load(os.path.join(G, "dbserver1500x8.npz"))
    X, Q = data(1500, 8, 15), data(8, 8, 16)
    off, nbr, nav = z["off"].astype(np.int64), z["nbr"].astype(np.int64), int(z["nav"])
    idc = np.arange(1500, dtype=np.int32)
    ix = amd.GpuIndex(8, 0)
    ix.attach_rows(X)
    ix.set_graph(off, nbr, nav)
    for name, spec in (("plain", None), ("lt700", ("<", 700)), ("ge1200", (">=", 1200))):
        if spec:
            ix.set_int_filter(idc, spec[0], spec[1])
        else:
            ix.set_int_filter(None, None, 0)
        for limit in (10, 100):
            ids, dist, cnt = ix.search(Q, limit, mode=amd.MODE_REFERENCE, intra_threads=1)
            for qi in range(len(Q)):
                want = z["%s_k%d_ids" % (name, limit)][qi]
                m = int((want >= 0).sum())
                assert int(cnt[qi]) == m, (name, limit, qi, cnt[qi], m)
                assert_topk_match(ids[qi, :m], dist[qi, :m], want[:m], z["%s_k%d_dist" % (name, limit)][qi][:m],
                                  what="%s k%d q%d" % (name, limit, qi))
    ix.set_int_filter(None, None, 0)
    ix.set_deleted(bitset(1500, range(0, 1500, 3)))
    ids, dist, cnt = ix.search(Q, 10, mode=amd.MODE_REFERENCE, intra_threads=1)
    for qi in range(len(Q)):
        want = z["deleted3_k10_ids"][qi]
        m = int((want >= 0).sum())
        assert int(cnt[qi]) == m
        assert_topk_match(ids[qi, :m], dist[qi, :m], want[:m], z["deleted3_k10_dist"][qi][:m])
    ix.close()


def test_known_answer_unit_circle_graph_plus_tail(amd, oracle):
    """DbServer.QueryDenseVectorDuringRebuild (db_server.cpp:1085-1245) scaled to N=2000: ids 0..499 in order."""
    N = 2000
    i = np.arange(N)
    X = np.stack([np.cos(np.pi * i / N), np.sin(np.pi * i / N)], 1).astype(np.float32)
    X = np.stack([oracle.normalize_insert(x) for x in X])
    q = oracle.normalize_query(np.array([1.0, 0.0], np.float32))
    off, nbr, nav = oracle.build_graph(1, X[:N // 2])
    ix = amd.GpuIndex(2, 1)
    ix.attach_rows(X)
    ix.set_graph(off, nbr, nav)
    for T in (1, 4):
        ids, dist, cnt = ix.search(q[None, :], 500, mode=amd.MODE_REFERENCE, intra_threads=T)
        assert cnt[0] == 500 and np.array_equal(ids[0], np.arange(500))
    ix.close()


def test_graph_file_roundtrip_with_reference_format(amd, oracle, tmp_path):
    z, off, nbr, nav = _golden_graph()
    X = data(2000, 32, 42)
    ix = amd.GpuIndex(32, 0)
    ix.attach_rows(X)
    ix.set_graph(off, nbr, nav)
    p = str(tmp_path / "ann_graph_1.bin")
    ix.save_graph(p)
    o2, n2, nav2, fid = oracle.graph_read(p)  # oracle reader was pinned against the reference's loader
    assert nav2 == nav and np.array_equal(o2, off) and np.array_equal(n2, nbr)
    ix2 = amd.GpuIndex(32, 0)
    ix2.attach_rows(X)
    ix2.load_graph(p)
    o3, n3, nav3 = ix2.get_graph()
    assert nav3 == nav and np.array_equal(o3, off) and np.array_equal(n3, nbr)


# ----------------------------------------------------------------------------------------------- properties at size
def test_flat_large_properties(amd):
    """200k x 128 on-device data: exactness against a float64 numpy scan for a query sample, sortedness,
    id-map arithmetic, device-pointer I/O (no host copies)."""
    import torch
    n, d, nq, k = 200_000, 128, 64, 10
    g = torch.Generator(device="cuda").manual_seed(42)
    X = torch.rand((n, d), generator=g, device="cuda", dtype=torch.float32)
    Qd = torch.rand((nq, d), generator=g, device="cuda", dtype=torch.float32)
    ix = amd.GpuIndex(d, 0).use_torch_stream()
    ix.attach_rows(X)
    ix.set_id_map(3, 8)
    ids = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    dist = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    cnt = torch.empty((nq,), dtype=torch.int32, device="cuda")
    ix.search(Qd, k, out=(ids, dist, cnt), mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
    ix.synchronize()
    Xh, Qh = X.cpu().numpy().astype(np.float64), Qd.cpu().numpy().astype(np.float64)
    ids_h, dist_h = ids.cpu().numpy(), dist.cpu().numpy()
    assert np.all(cnt.cpu().numpy() == k) and np.all(np.diff(dist_h, axis=1) >= 0)
    for qi in range(0, nq, 7):
        ex = ((Xh - Qh[qi]) ** 2).sum(1)
        o = np.argsort(ex, kind="stable")[:k]
        assert_topk_match((ids_h[qi] - 3) // 8, dist_h[qi], o, ex[o])
        assert np.all((ids_h[qi] - 3) % 8 == 0)
    ix.close()


def test_merge_topk(amd):
    rng = np.random.default_rng(0)
    S, nq, k = 4, 9, 10
    d = np.sort(rng.random((S, nq, k), dtype=np.float32), axis=2)
    ids = rng.integers(0, 10 ** 9, (S, nq, k)).astype(np.int64)
    d[1, :, 7:] = np.inf
    ids[1, :, 7:] = -1
    od = np.empty((nq, k), np.float32)
    oi = np.empty((nq, k), np.int64)
    amd.merge_topk(d, ids, od, oi)
    for q in range(nq):
        pairs = sorted((float(d[s, q, e]), int(ids[s, q, e])) for s in range(S) for e in range(k) if ids[s, q, e] >= 0)[:k]
        assert [p[1] for p in pairs] == list(oi[q]) and np.allclose([p[0] for p in pairs], od[q])


# ----------------------------------------------------------------------------------------------- MFMA filter engine
@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("n,d,nq", [(70_000, 768, 40), (100_000, 128, 130), (66_000, 100, 64), (80_000, 33, 33)])
def test_mfma_engine_is_exact(amd, oracle, metric, n, d, nq):
    """fp16-MFMA lower-bound filter + exact fp32 re-rank must return exactly what the fp32 stream scan returns
    (same rows, same distance bits), and match the CPU oracle on a query sample."""
    X = data(n, d, 7 + d)
    Q = data(nq, d, 8 + d)
    if metric == 1:
        X = amd.normalize_rows(X, only_if_nonzero=True)
        Q = amd.normalize_rows(Q, only_if_nonzero=False)
    ix = amd.GpuIndex(d, metric)
    ix.attach_rows(X)
    for k in (1, 10, 100):
        a = ix.search(Q, k, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA)
        st = ix.stats()
        b = ix.search(Q, k, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
        assert st["overflow_queries"] == 0 and st["rerank_rows"] > 0
        # expected candidates ~ k * (rows / rows already ranked) summed over the stages; allow 2x
        assert st["rerank_rows"] < nq * (2.0 * k * n / 4096 + 64), "filter is not selective: %d" % st["rerank_rows"]
        assert np.array_equal(a[0], b[0]), "k=%d: %d rows differ" % (k, (a[0] != b[0]).sum())
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    for qi in range(0, nq, 13):
        rid, rd = oracle.topk_flat(metric, X, Q[qi], 10)
        ids, dist, cnt = ix.search(Q[qi:qi + 1].repeat(32, 0), 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA)
        assert_topk_match(ids[5], dist[5], rid, rd, what="mfma vs oracle q%d" % qi)
    ix.close()


@pytest.mark.parametrize("n,d,nq,metric", [(150_000, 256, 300, 0), (150_000, 384, 1100, 2), (100_000, 1024, 64, 0),
                                           (70_000, 256, 8448, 0), (20_000, 512, 96, 1)])
def test_mfma_v7_shapes_are_exact(amd, n, d, nq, metric):
    """The persistent 4-wavefront kernel over its geometry corners: the minimum K depth (d_pad 256 = 4 K-steps), an odd
    number of step pairs (384), a deep K (1024), several query tiles incl. a padded last one (300, 1100 queries), more
    query tiles than workgroups per XCD (8448 queries: every workgroup walks two query tiles and reloads its thresholds),
    and a table barely above the seeded-staging cut-over.  MFMA engine == fp32 stream engine, bit for bit."""
    X = data(n, d, 70 + d)
    Q = data(nq, d, 71 + d)
    if metric == 1:
        X = amd.normalize_rows(X, only_if_nonzero=True)
        Q = amd.normalize_rows(Q, only_if_nonzero=False)
    ix = amd.GpuIndex(d, metric)
    ix.attach_rows(X)
    for k in (10, 64):
        a = ix.search(Q, k, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA)
        st = ix.stats()
        assert st["overflow_queries"] == 0 and st["rerank_rows"] > 0
        b = ix.search(Q, k, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
        assert np.array_equal(a[0], b[0]), "k=%d: %d rows differ" % (k, (a[0] != b[0]).sum())
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    ix.close()


@pytest.mark.parametrize("env", [{"EPS_MFMA_MAX_BATCH": "16384"}, {"EPS_MFMA_SYNC_SHIFT": "0"}, {"EPS_MFMA_SYNC_SHIFT": "5"},
                                 {"EPS_MFMA_GROUPSYNC": "0"}, {"EPS_MFMA_SEED": "0"}])
def test_mfma_v7_tile_walk_variants_are_exact(amd, monkeypatch, env):
    """The filter kernel's tile walk under its host-side knobs (read per search call): one pass over 8448 queries instead
    of slices of 2048 (33 query tiles > 32 workgroups per XCD: some workgroups walk two query tiles, step their (row,
    query) tile counters through a wrap, reload thresholds per tile and do not rendezvous), the group rendezvous before
    every tile / every 32nd tile / never, and unseeded staging (loose first-stage thresholds: thousands of hits per
    wavefront, so the pending-hit list in LDS fills, flushes and overflows to the direct path).  k = 200 keeps the lists
    busy in every stage.  MFMA engine == fp32 stream engine, bit for bit."""
    for kk, vv in env.items():
        monkeypatch.setenv(kk, vv)
    n, d, nq = 70_000, 256, 8448 if "EPS_MFMA_MAX_BATCH" in env else 600
    X = data(n, d, 170 + d)
    Q = data(nq, d, 171 + d)
    ix = amd.GpuIndex(d, 0)
    ix.attach_rows(X)
    for k in (10, 200):
        a = ix.search(Q, k, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA)
        st = ix.stats()
        assert st["overflow_queries"] == 0 and st["rerank_rows"] > 0
        b = ix.search(Q, k, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
        assert np.array_equal(a[0], b[0]), "k=%d: %d rows differ" % (k, (a[0] != b[0]).sum())
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    ix.close()


@pytest.mark.parametrize("nq", [1, 3, 8, 17, 31, 100, 128, 129])
def test_small_batches_auto_engine_is_exact(amd, nq):
    """FLAT_AUTO is a cost decision between two engines that return the same bits: from 8 queries on (or earlier, once
    the fp16 mirror exists) a big table goes through the MFMA filter.  AUTO == MFMA == stream, for every small batch."""
    n, d = 120_000, 256
    X = data(n, d, 90)
    Q = data(nq, d, 91 + nq)
    ix = amd.GpuIndex(d, 0)
    ix.attach_rows(X)
    ref = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
    for eng in (amd.FLAT_AUTO, amd.FLAT_MFMA, amd.FLAT_AUTO):
        got = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=eng)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])
    ix.close()


def test_mfma_engine_with_deleted_filter_and_ties(amd, oracle):
    n, d, nq = 90_000, 64, 48
    rng = np.random.default_rng(3)
    base = rng.random((300, d), dtype=np.float32)
    X = base[rng.integers(0, 300, n)]                     # every row has ~300 exact duplicates: massive ties
    Q = rng.random((nq, d), dtype=np.float32)
    bits = bitset(n, range(0, n, 3))
    idc = np.arange(n, dtype=np.int64)
    ix = amd.GpuIndex(d, 0)
    ix.attach_rows(X)
    ix.set_deleted(bits)
    ix.set_int_filter(idc, ">=", 1000)
    for k in (10, 64):
        a = ix.search(Q, k, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA)
        b = ix.search(Q, k, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    flt, keep = make_filter(deleted=bits, attr=idc, stride=8, width=8, op=">=", value=1000)
    rid, rd = oracle.topk_flat(0, X, Q[0], 10, flt=flt)
    assert np.array_equal(a[0][0][:10], rid)               # ties resolve by id exactly as Candidate::operator<
    ix.close()


def test_mfma_engine_adversarial_order_falls_back_exactly(amd):
    """Rows sorted from far to near: every stage finds better rows than the threshold assumed, the candidate
    buffers overflow, and the engine must fall back to the exact scan rather than drop results."""
    import torch
    n, d, nq = 300_000, 32, 64
    g = torch.Generator(device="cuda").manual_seed(1)
    q0 = torch.rand((1, d), generator=g, device="cuda")
    X = torch.rand((n, d), generator=g, device="cuda")
    order = torch.argsort(((X - q0) ** 2).sum(1), descending=True)
    X = X[order].contiguous()
    Q = (q0 + 0.01 * torch.rand((nq, d), generator=g, device="cuda")).contiguous()
    ix = amd.GpuIndex(d, 0).use_torch_stream()
    ix.attach_rows(X)
    outs = []
    for eng in (amd.FLAT_MFMA, amd.FLAT_STREAM):
        ids = torch.empty((nq, 10), dtype=torch.int64, device="cuda")
        dist = torch.empty((nq, 10), dtype=torch.float32, device="cuda")
        cnt = torch.empty((nq,), dtype=torch.int32, device="cuda")
        ix.search(Q, 10, out=(ids, dist, cnt), mode=amd.MODE_FLAT, flat_engine=eng)
        ix.synchronize()
        outs.append((ids.cpu().numpy(), dist.cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    ix.close()


def test_mfma_engine_selective_filter_retries_with_more_slots(amd):
    """99.5 % of the rows deleted at random: thresholds come from visible rows only, so ~200 x more rows pass the bound than
    for an unfiltered search and the first pass overflows; the engine retries with 16 x the candidate slots (still far
    cheaper than the stream scan) and must return exactly the stream engine's answer."""
    n, d, nq = 400_000, 64, 96
    rng = np.random.default_rng(11)
    X, Q = data(n, d, 120), data(nq, d, 121)
    dele = rng.random(n) < 0.995
    ix = amd.GpuIndex(d, 0)
    ix.attach_rows(X)
    ix.set_deleted(np.packbits(dele, bitorder="little"))
    a = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA)
    st = ix.stats()
    b = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
    assert st["overflow_queries"] > 0                      # the 4096-slot pass did overflow ...
    assert st["rerank_rows"] > 4096 * 8                    # ... and the retry re-ranked long lists instead of scanning
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert not dele[a[0][a[0] >= 0]].any()
    ix.close()


@pytest.mark.parametrize("d", [64, 256])
def test_mfma_engine_positional_filter_hiding_the_head(amd, d):
    """"Only the newest rows": the first two thirds of the table are invisible.  Seeds come from a sample spread over the
    whole table, so thresholds exist from the start and nothing overflows; answer == stream engine."""
    n, nq = 300_000, 80
    X, Q = data(n, d, 130), data(nq, d, 131)
    dele = np.zeros(n, bool)
    dele[:200_000] = True
    ix = amd.GpuIndex(d, 0)
    ix.attach_rows(X)
    ix.set_deleted(np.packbits(dele, bitorder="little"))
    a = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA)
    st = ix.stats()
    b = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
    # thresholds stay at the seeds' level until the scan reaches visible rows, so some queries still need the 16 x retry;
    # without sampled seeds EVERY query overflows in every stage and the batch ends on the stream engine
    assert st["overflow_queries"] < nq and st["rerank_rows"] > 0   # (every query in every stage, then the stream engine, without them)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert (a[0] >= 200_000).all()
    ix.close()


def test_graph_search_csr_fallback_for_high_degree_nodes(amd, oracle):
    """Connectivity repair can give a node any out-degree; above 64 the device keeps the CSR form instead of the
    fixed-stride adjacency.  Same answer as the oracle either way, duplicates in an adjacency list included."""
    z, off, nbr, nav = _golden_graph()
    X, Q = data(2000, 32, 42), data(8, 32, 45)
    lists = [list(nbr[off[i]:off[i + 1]]) for i in range(2000)]
    lists[nav] += list(range(100, 190)) + [lists[nav][0]]        # 90 extra edges + one duplicate
    lists[7] += [3, 3, 3]
    off2 = np.zeros(2001, np.int64)
    off2[1:] = np.cumsum([len(l) for l in lists])
    nbr2 = np.concatenate([np.asarray(l, np.int64) for l in lists])
    ix = amd.GpuIndex(32, 0)
    ix.attach_rows(X)
    ix.set_graph(off2, nbr2, nav)
    ids, dist, cnt = ix.search(Q, 50, mode=amd.MODE_GRAPH, intra_threads=1)
    init = oracle.prepare_init_ids(off2, nbr2, nav, 500)
    for qi, q in enumerate(Q):
        oid, od, _ = oracle.search_impl(0, X, off2, nbr2, init, q, T=1, L=500)
        assert_topk_match(ids[qi], dist[qi], oid[:50], od[:50], what="csr q%d" % qi)
    ix.close()


def test_graph_search_large_batch_slices(amd, oracle):
    """1500 queries in one call (more workgroups than fit at once, visited bitmaps of the whole batch)."""
    z, off, nbr, nav = _golden_graph()
    X = data(2000, 32, 42)
    Q = np.tile(data(16, 32, 43), (94, 1))[:1500]
    ix = amd.GpuIndex(32, 0)
    ix.attach_rows(X)
    ix.set_graph(off, nbr, nav)
    ids, dist, cnt = ix.search(Q, 10, mode=amd.MODE_GRAPH, intra_threads=1)
    for qi in range(1500):
        assert_topk_match(ids[qi], dist[qi], z["ids_m0"][qi % 16][:10], z["dist_m0"][qi % 16][:10], what="q%d" % qi)
    ix.close()


# ----------------------------------------------------------------------------------------------- edge cases
def test_edge_empty_and_tiny_tables(amd, oracle):
    ix = amd.GpuIndex(8, 0)
    ix.attach_rows(np.zeros((0, 8), np.float32))
    ids, dist, cnt = ix.search(data(3, 8, 1), 5)
    assert list(cnt) == [0, 0, 0] and np.all(ids == -1) and np.all(np.isinf(dist))
    X = data(3, 8, 2)
    ix.attach_rows(X)
    ids, dist, cnt = ix.search(data(2, 8, 3), 10)             # k > n
    assert list(cnt) == [3, 3] and np.all(ids[:, 3:] == -1)
    ix.set_deleted(np.array([0b111], np.uint8))               # everything deleted
    ids, dist, cnt = ix.search(data(2, 8, 3), 10)
    assert list(cnt) == [0, 0]
    ix.set_deleted(None)
    ix.set_int_filter(np.arange(3, dtype=np.int64), ">", 100)  # filter passes nothing
    assert list(ix.search(data(2, 8, 3), 10)[2]) == [0, 0]
    ix.close()


@pytest.mark.parametrize("d", [1, 3, 5, 2048, 4096, 8192])
def test_edge_dimensions(amd, oracle, d):
    n = 300 if d > 1000 else 2000
    X = data(n, d, d) - 0.5                                     # negative values too
    Q = data(3, d, d + 1) - 0.5
    for metric in (0, 2):
        ix = amd.GpuIndex(d, metric)
        ix.attach_rows(X)
        ids, dist, cnt = ix.search(Q, 7, mode=amd.MODE_FLAT)
        for qi in range(3):
            rid, rd = oracle.topk_flat(metric, X, Q[qi], 7)
            assert_topk_match(ids[qi], dist[qi], rid, rd, rtol=2e-4, atol=2e-5, what="d=%d m=%d" % (d, metric))
        ix.close()


def test_edge_nan_query_does_not_hang(amd):
    """A zero COSINE query normalises to NaN in the reference (vector.cpp:60-69); the scan must still terminate."""
    X = data(5000, 16, 1)
    q = amd.normalize_rows(np.zeros((1, 16), np.float32), only_if_nonzero=False)
    assert np.isnan(q).all()
    ix = amd.GpuIndex(16, 1)
    ix.attach_rows(X)
    ids, dist, cnt = ix.search(q, 5, mode=amd.MODE_FLAT)
    assert cnt[0] in (0, 5)
    ix.close()


def test_append_rows_and_tail_search(amd, oracle):
    """Rows inserted after the last rebuild are searched through the brute-force tail (:885-900): graph over the first
    1000 rows, two appends, result must equal the oracle's Search() on the same graph."""
    z = np.load(os.path.join(G, "dbserver1500x8.npz"))
    off, nbr, nav = z["off"].astype(np.int64), z["nbr"].astype(np.int64), int(z["nav"])
    X, Q = data(1500, 8, 15), data(8, 8, 16)
    ix = amd.GpuIndex(8, 0)
    ix.attach_rows(X[:1000])
    ix.set_graph(off, nbr, nav)
    ix.append_rows(X[1000:1200])
    ix.append_rows(X[1200:])
    assert ix.row_count == 1500
    ids, dist, cnt = ix.search(Q, 10, mode=amd.MODE_REFERENCE, intra_threads=1)
    for qi in range(8):
        assert_topk_match(ids[qi], dist[qi], z["plain_k10_ids"][qi], z["plain_k10_dist"][qi])
    ix.close()


def test_append_extends_the_fp16_mirror_incrementally(amd, oracle):
    """appended rows are converted into the fp16 mirror on their own (the mirror of the rows already there is
    kept); results after the append equal the exact stream engine and the oracle, including hits in the appended tail."""
    n0, n1, d = 70_000, 9_000, 128
    X, Q = data(n0 + n1, d, 7), data(24, d, 8)
    X[n0 + 5] = Q[0] + 1e-3      # planted neighbours inside the appended rows
    X[n0 + n1 - 1] = Q[1] - 2e-3
    ix = amd.GpuIndex(d, 0)
    ix.attach_rows(X[:n0])
    a = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA)
    assert ix.stats()["rerank_rows"] > 0
    ix.append_rows(X[n0:n0 + 4000])
    ix.append_rows(X[n0 + 4000:])
    assert ix.row_count == n0 + n1
    b = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA)
    assert ix.stats()["rerank_rows"] > 0
    c = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
    assert np.array_equal(b[0], c[0]) and np.array_equal(b[1], c[1])
    assert b[0][0][0] == n0 + 5 and b[0][1][0] == n0 + n1 - 1
    for qi in range(0, len(Q), 5):
        rid, rd = oracle.topk_flat(0, X, Q[qi], 10)
        assert_topk_match(b[0][qi], b[1][qi], rid, rd, what="append q%d" % qi)
    ix.close()


def test_attach_rows_revalidates_dependent_state(amd):
    """re-attaching fewer rows drops a graph that covers more rows than are attached and a bitset / filter column of
    the old length; appending rows makes a too-short bitset an error instead of an out-of-bounds read."""
    z, off, nbr, nav = _golden_graph()
    X = data(2000, 32, 42)
    ix = amd.GpuIndex(32, 0)
    ix.attach_rows(X)
    ix.set_graph(off, nbr, nav)
    ix.set_deleted(bitset(2000, [1, 2, 3]))
    ix.attach_rows(X[:900])
    assert ix.graph_info()[0] == 0
    with pytest.raises(amd.EpsillaError):
        ix.search(X[:1], 5, mode=amd.MODE_GRAPH)
    ids, _, _ = ix.search(X[1:2], 1, mode=amd.MODE_FLAT)   # the old bitset (row 1 deleted) is still long enough and still applies
    assert ids[0][0] != 1
    ix.set_deleted(bitset(900, [1]))
    ix.append_rows(X[900:1000])
    with pytest.raises(amd.EpsillaError):
        ix.search(X[:1], 5, mode=amd.MODE_FLAT)
    ix.set_deleted(None)
    ids, _, _ = ix.search(X[950:951], 1, mode=amd.MODE_FLAT)
    assert ids[0][0] == 950
    # a one-row and an empty build (ADVICE: build() with n == 1 failed)
    ix1 = amd.GpuIndex(32, 0)
    ix1.attach_rows(X[:1])
    ix1.build(1)
    assert ix1.graph_info()[:2] == (1, 0)
    ix.close()
    ix1.close()


def test_mfma_filter_is_exact_at_high_dimension_with_cancellation(amd):
    """the fp32 slack of the filter threshold has to grow with d.  d = 8192, rows and queries all within 1e-3 of one
    point (distances are ~1e-5 of the squared norms: massive cancellation in |x|^2 - 2 q.x + |q|^2): the MFMA engine must still
    return exactly what the fp32 stream engine returns."""
    import torch
    n, d, nq = 66_000, 8192, 16
    g = torch.Generator(device="cuda").manual_seed(5)
    c = torch.rand((1, d), generator=g, device="cuda")
    X = c + 1e-3 * torch.randn((n, d), generator=g, device="cuda")
    Qd = c + 1e-3 * torch.randn((nq, d), generator=g, device="cuda")
    ix = amd.GpuIndex(d, 0).use_torch_stream()
    ix.attach_rows(X)
    outs = []
    for eng in (amd.FLAT_MFMA, amd.FLAT_STREAM):
        o = (torch.empty((nq, 10), dtype=torch.int64, device="cuda"), torch.empty((nq, 10), dtype=torch.float32, device="cuda"),
             torch.empty((nq,), dtype=torch.int32, device="cuda"))
        ix.search(Qd, 10, out=o, mode=amd.MODE_FLAT, flat_engine=eng)
        ix.synchronize()
        outs.append((o[0].cpu().numpy(), o[1].cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1], outs[1][1])
    ix.close()


def test_load_table_segment_file_written_by_the_reference(amd, tmp_path):
    """`data_mvp.bin` written by the reference's own TableSegmentMVP::SaveTableSegment (through its DBServer:
    INT pk, STRING, FLOAT attribute, two dense fields, deletes) is read straight into HBM by eps_index_load_table; flat searches on
    the loaded field - plain, with the file's deleted bitset, with a filter program over the file's attribute rows - return what the
    reference DBServer returns."""
    import glob
    from oracle.pyoracle import Ref, ref_available
    if not ref_available():
        pytest.skip("needs oracle/_ref")
    ref = Ref()
    n = 900
    X8, X16 = data(n, 8, 51), data(n, 16, 52)
    price = np.random.default_rng(53).random(n)
    schema = {"name": "T", "fields": [{"name": "ID", "dataType": "INT", "primaryKey": True}, {"name": "Tag", "dataType": "STRING"},
                                       {"name": "Price", "dataType": "FLOAT"},
                                       {"name": "A", "dataType": "VECTOR_FLOAT", "dimensions": 8, "metricType": "EUCLIDEAN"},
                                       {"name": "B", "dataType": "VECTOR_FLOAT", "dimensions": 16, "metricType": "EUCLIDEAN"}]}
    recs = [{"ID": int(i), "Tag": "tag-%d" % (i * 37 % 101), "Price": float(np.float32(price[i])), "A": [float(x) for x in X8[i]],
             "B": [float(x) for x in X16[i]]} for i in range(n)]
    ref.L.ref_config(1, 500, 1, 0, 2)
    db = ref.db(str(tmp_path / "db"))
    assert db.create_table(schema) == 0 and db.insert("T", recs) == 0
    assert db.delete("T", [5, 6, 7, 400]) == 0
    assert db.rebuild() == 0                      # Rebuild() saves the table segment (db_mvp.cpp / table_mvp.cpp)
    files = glob.glob(str(tmp_path / "db" / "*" / "data_mvp.bin"))
    assert len(files) == 1, files
    Q = data(6, 16, 54)
    ix = amd.GpuIndex(16, 0)
    assert ix.load_table(files[0], primitive_offset=8, var_len_attrs=1, dense_dims=[8, 16], field=1) == n
    for flt, prog in (("", None), ("Price < 0.4 AND ID >= 100", [("f32", 4), ("const", 0.4), ("<",), ("i32", 0), ("const", 100), (">=",), ("and",)])):
        ix.set_filter_program(prog)               # rows = None: the attribute rows the loader kept
        ids, dist, cnt = ix.search(Q, 10, mode=amd.MODE_FLAT)
        for qi, q in enumerate(Q):
            rc, res = db.search("T", "B", q, 10, fields=("ID",), flt=flt)
            assert rc == 0
            assert [r["ID"] for r in res] == list(ids[qi][:cnt[qi]]), (flt, qi)      # ID == row index here
            assert np.allclose([r["@distance"] for r in res], dist[qi][:cnt[qi]], rtol=1e-4)
    ix8 = amd.GpuIndex(8, 0)
    assert ix8.load_table(files[0], primitive_offset=8, var_len_attrs=1, dense_dims=[8, 16], field=0) == n
    ids, _, _ = ix8.search(X8[10:11], 1, mode=amd.MODE_FLAT)
    assert ids[0][0] == 10
    with pytest.raises(amd.EpsillaError):
        ix8.load_table(files[0], primitive_offset=8, var_len_attrs=3, dense_dims=[8, 16], field=0)   # wrong layout: detected, not read out of bounds
    db.close()
    ref.L.ref_config(4, 500, 1, 0, 4)
    ix.close()
    ix8.close()
