"""Pins the plain-C oracle (oracle/epsilla_oracle.c) against the reference itself, compiled verbatim
(oracle/_ref, built from /root/reference by oracle/Makefile).  CPU only.  Skipped where the reference
build is absent; tests/test_golden.py covers the same ground from committed fixtures."""
import numpy as np
import pytest

pytestmark = pytest.mark.ref


def data(n, d, seed=42):
    rng = np.random.default_rng(seed)
    return rng.random((n, d), dtype=np.float32)


@pytest.mark.parametrize("d", [1, 2, 3, 4, 5, 7, 8, 31, 32, 33, 100, 128, 768, 960, 1536])
def test_distance_bit_exact(oracle, ref, d):
    X = data(64, d, 1)
    Q = data(4, d, 2) * 2 - 0.5
    for q in Q:
        for x in X:
            assert oracle.l2sqr(x, q) == ref.l2sqr(x, q)
            assert oracle.ip(x, q) == ref.ip(x, q)
            for m in (0, 1, 2):
                assert oracle.dist(m, x, q) == ref.dist(m, x, q)


def test_normalize(oracle, ref):
    for d in (2, 4, 33, 768):
        v = data(1, d, 3)[0]
        assert np.array_equal(oracle.normalize_query(v), ref.normalize(v))


@pytest.fixture(scope="module")
def small(ref):
    X = data(2000, 32, 42)
    g = ref.build_graph(X, metric=0, threads=1)
    off, nbr, nav = ref.graph_arrays(g)
    return X, g, off, nbr, nav


def test_prepare_init_ids(oracle, ref, small):
    X, g, off, nbr, nav = small
    for L in (500, 777):
        ex = ref.executor(g, X, T=1, L=L)
        assert np.array_equal(ref.init_ids(ex, L), oracle.prepare_init_ids(off, nbr, nav, L))


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("L", [500, 600])
def test_search_impl_T1_bit_exact(oracle, ref, small, metric, L):
    """T = 1 is deterministic on both sides: the whole master queue must agree, ids and distances."""
    X, g, off, nbr, nav = small
    Q = data(16, X.shape[1], 43)
    ex = ref.executor(g, X, metric=metric, T=1, L=L)
    init = oracle.prepare_init_ids(off, nbr, nav, L)
    for q in Q:
        rid, rd = ref.search_impl(ex, q, L)
        oid, od, _ = oracle.search_impl(metric, X, off, nbr, init, q, T=1, L=L)
        assert np.array_equal(rid, oid)
        assert np.array_equal(rd, od)


def test_search_impl_T4_same_answer(oracle, ref, small):
    """T = 4 is racy in the reference; both sides must still converge to the same top-10 here."""
    X, g, off, nbr, nav = small
    Q = data(16, X.shape[1], 44)
    ex = ref.executor(g, X, metric=0, T=4, L=500)
    init = oracle.prepare_init_ids(off, nbr, nav, 500)
    for q in Q:
        rid, rd = ref.search_impl(ex, q, 10)
        oid, od, _ = oracle.search_impl(0, X, off, nbr, init, q, T=4, L=500)
        assert np.array_equal(rid, oid[:10])
        assert np.array_equal(rd, od[:10])


def test_dist_eval_count_T1(oracle, ref, small):
    X, g, off, nbr, nav = small
    q = data(1, X.shape[1], 45)[0]
    ex = ref.executor(g, X, metric=0, T=1, L=500, count=True)
    ref.L.ref_dist_calls_reset()
    ref.search_impl(ex, q, 10)
    n_ref = ref.L.ref_dist_calls_reset()
    init = oracle.prepare_init_ids(off, nbr, nav, 500)
    _, _, n_or = oracle.search_impl(0, X, off, nbr, init, q, T=1, L=500)
    assert n_ref == n_or


@pytest.mark.parametrize("n,d", [(600, 16), (1500, 24)])
def test_nsg_from_same_knn_bit_exact(oracle, ref, n, d):
    """Feed the reference's NsgIndex and the restatement the same kNN graph: identical CSR + nav."""
    X = data(n, d, 7)
    knn = oracle.knn_exact(0, X, 100)
    g = ref.nsg_from_knn(X, knn, threads=1, seed=100)
    roff, rnbr, rnav = ref.graph_arrays(g)
    ooff, onbr, onav = oracle.nsg_build(X, knn, seed=100)
    assert rnav == onav
    assert np.array_equal(roff, ooff)
    assert np.array_equal(rnbr, onbr)


def test_nsg_from_nndescent_knn(oracle, ref):
    """Same, but on the (approximate, ragged) kNN lists NN-Descent really produces."""
    X = data(800, 16, 9)
    knn = ref.knn_graph(X, K=100, metric=0, threads=1)
    g = ref.nsg_from_knn(X, knn, threads=1, seed=100)
    roff, rnbr, rnav = ref.graph_arrays(g)
    ooff, onbr, onav = oracle.nsg_build(X, knn, seed=100)
    assert rnav == onav and np.array_equal(roff, ooff) and np.array_equal(rnbr, onbr)


def test_knn_exact_contains_nndescent(oracle, ref):
    """NN-Descent approximates the exact list the oracle (and the GPU build) computes."""
    X = data(1000, 16, 11)
    approx = ref.knn_graph(X, K=100, metric=0, threads=1)
    exact = oracle.knn_exact(0, X, 100)
    hit = sum(len(set(a[a >= 0]) & set(e)) for a, e in zip(approx, exact))
    assert hit / exact.size > 0.95


def test_graph_file_format_roundtrip(oracle, ref, small, tmp_path):
    X, g, off, nbr, nav = small
    # reference writes, oracle reads
    (tmp_path / "7").mkdir()
    assert ref.L.ref_graph_save(g, str(tmp_path).encode(), 7, 3) == 0
    o2, n2, nav2, fid = oracle.graph_read(str(tmp_path / "7" / "ann_graph_3.bin"))
    assert nav2 == nav and fid == 0 and np.array_equal(o2, off) and np.array_equal(n2, nbr)
    # oracle writes, reference reads
    (tmp_path / "8").mkdir()
    assert oracle.graph_write(str(tmp_path / "8" / "ann_graph_1.bin"), off, nbr, nav) == 0
    g2 = ref.L.ref_graph_load(str(tmp_path).encode(), 8, 1)
    assert g2
    o3, n3, nav3 = ref.graph_arrays(g2)
    assert nav3 == nav and np.array_equal(o3, off) and np.array_equal(n3, nbr)


CITY_SCHEMA = {
    "name": "MyTable",
    "fields": [
        {"name": "ID", "dataType": "INT", "primaryKey": True},
        {"name": "Doc", "dataType": "STRING"},
        {"name": "EmbeddingEuclidean", "dataType": "VECTOR_FLOAT", "dimensions": 4, "metricType": "EUCLIDEAN"},
        {"name": "EmbeddingDotProduct", "dataType": "VECTOR_FLOAT", "dimensions": 4, "metricType": "DOT_PRODUCT"},
        {"name": "EmbeddingCosine", "dataType": "VECTOR_FLOAT", "dimensions": 4, "metricType": "COSINE"},
    ],
}
CITIES = [
    (1, "Berlin", [0.05, 0.61, 0.76, 0.74]),
    (2, "London", [0.19, 0.81, 0.75, 0.11]),
    (3, "Moscow", [0.36, 0.55, 0.47, 0.94]),
    (4, "San Francisco", [0.18, 0.01, 0.85, 0.80]),
    (5, "Shanghai", [0.24, 0.18, 0.22, 0.44]),
]


def test_dbserver_flat_matches_oracle(oracle, ref, tmp_path):
    """Through the reference's DBServer (JSON insert -> BruteForceSearch) vs. the oracle on raw floats."""
    db = ref.db(str(tmp_path / "db"))
    assert db.create_table(CITY_SCHEMA) == 0
    recs = [{"ID": i, "Doc": c, "EmbeddingEuclidean": v, "EmbeddingDotProduct": v, "EmbeddingCosine": v}
            for i, c, v in CITIES]
    assert db.insert("MyTable", recs) == 0
    q = np.array([0.35, 0.55, 0.47, 0.94], np.float32)
    X = np.array([v for _, _, v in CITIES], np.float32)
    Xn = np.stack([oracle.normalize_insert(x) for x in X])
    for field, metric, rows, qq in (("EmbeddingEuclidean", 0, X, q), ("EmbeddingDotProduct", 2, X, q),
                                    ("EmbeddingCosine", 1, Xn, oracle.normalize_query(q))):
        rc, res = db.search("MyTable", field, q, 6, fields=("ID", "Doc"))
        assert rc == 0
        ids, ds = oracle.topk_flat(metric, rows, qq, 6)
        assert [r["ID"] for r in res] == [int(i) + 1 for i in ids]
        assert np.allclose([r["@distance"] for r in res], ds.astype(np.float64), rtol=0, atol=0)
    # deleted bitset honoured (DbServer.DeleteByPK)
    assert db.delete("MyTable", [1, 2, 3, 4]) == 0
    rc, res = db.search("MyTable", "EmbeddingEuclidean", q, 6, fields=("ID", "Doc"))
    from oracle.pyoracle import make_filter
    flt, keep = make_filter(deleted=np.array([0b00001111], np.uint8))
    ids, ds = oracle.topk_flat(0, X, q, 6, flt=flt)
    assert [r["Doc"] for r in res] == ["Shanghai"] and list(ids) == [4]
    db.close()


def test_dbserver_filter_matches_oracle(oracle, ref, tmp_path):
    """`ID <= 2` style int filters (DbServer.DenseVectorFilter, db_server.cpp:1407-1630)."""
    from oracle.pyoracle import make_filter
    db = ref.db(str(tmp_path / "db"))
    schema = {"name": "T", "fields": [{"name": "ID", "dataType": "INT", "primaryKey": True},
                                       {"name": "V", "dataType": "VECTOR_FLOAT", "dimensions": 8,
                                        "metricType": "EUCLIDEAN"}]}
    assert db.create_table(schema) == 0
    X = data(300, 8, 5)
    assert db.insert("T", [{"ID": int(i), "V": [float(x) for x in X[i]]} for i in range(300)]) == 0
    q = data(1, 8, 6)[0]
    ids_col = np.arange(300, dtype=np.int32)
    for op, val in (("<", 150), ("<=", 2), (">=", 290), ("=", 17), (">", 100)):
        rc, res = db.search("T", "V", q, 10, flt="ID %s %d" % (op, val))
        assert rc == 0, res
        flt, keep = make_filter(attr=ids_col, stride=4, width=4, op=op, value=val)
        ids, ds = oracle.topk_flat(0, X, q, 10, flt=flt)
        assert [r["ID"] for r in res] == [int(i) for i in ids]
        assert [r["@distance"] for r in res] == [float(x) for x in ds]
    db.close()


def test_dbserver_graph_plus_tail_matches_oracle(oracle, ref, tmp_path):
    """Rebuild on the first half, insert the rest: graph search + brute-force tail + merge
    (Search, vec_search_executor.cpp:833-935) — the situation DbServer.QueryDenseVectorDuringRebuild pins."""
    ref.L.ref_config(1, 500, 1, 0, 1)
    db = ref.db(str(tmp_path / "db"))
    schema = {"name": "T", "fields": [{"name": "ID", "dataType": "INT", "primaryKey": True},
                                       {"name": "V", "dataType": "VECTOR_FLOAT", "dimensions": 8,
                                        "metricType": "EUCLIDEAN"}]}
    assert db.create_table(schema) == 0
    X = data(1500, 8, 15)
    recs = [{"ID": int(i), "V": [float(x) for x in X[i]]} for i in range(1500)]
    assert db.insert("T", recs[:1000]) == 0
    assert db.rebuild() == 0
    assert db.insert("T", recs[1000:]) == 0
    # same graph for the oracle: read the file the reference saved
    off, nbr, nav, _ = oracle.graph_read(str(tmp_path / "db" / "0" / "ann_graph_1.bin"))
    assert len(off) - 1 == 1000
    Q = data(8, 8, 16)
    for q in Q:
        for limit in (10, 100):
            rc, res = db.search("T", "V", q, limit)
            assert rc == 0
            ids, ds, _ = oracle.search(0, X, 1000, off, nbr, nav, q, limit, T=1, L=500)
            assert [r["ID"] for r in res] == [int(i) for i in ids]
            assert [r["@distance"] for r in res] == [float(x) for x in ds]
    ref.L.ref_config(4, 500, 1, 0, 16)
    db.close()


def test_select_edge_bit_exact(oracle, ref):
    """SyncPrune's sort + SelectEdge (nsg.cpp:557-567, 655-685) on identical candidate pools: the restatement equals the
    reference's own (protected) member, with and without the candidate_pool_size limit and for small out-degrees."""
    rng = np.random.default_rng(3)
    X = rng.random((4000, 24), dtype=np.float32)
    for node in range(0, 400, 11):
        cands = rng.choice(4000, size=420, replace=False).astype(np.int64)
        cands[0] = node
        for depth, R in ((300, 50), (0, 50), (300, 8), (40, 50)):
            a = oracle.select_edge(X, node, cands, depth, R)
            b = ref.select_edge(X, node, cands, depth, R)
            assert np.array_equal(a, b), (node, depth, R)


def test_inter_insert_bit_exact(oracle, ref):
    """InterInsert over all nodes in order (nsg.cpp:531-536, 583-653) on identical edge lists: the restatement equals the
    reference's own (protected) member called node by node - including its quirk that a re-selected list keeps its stale tail
    (nsg.cpp:632-639) - for out-degrees small enough that most lists overflow and large enough that none does."""
    rng = np.random.default_rng(5)
    n, d = 2500, 16
    X = rng.random((n, d), dtype=np.float32)
    knn = oracle.knn_exact(0, X, 30)
    for R in (8, 16, 50):
        ids = np.full((n, R), -1, np.int64)
        deg = np.zeros(n, np.int64)
        for v in range(n):
            e = oracle.select_edge(X, v, knn[v], 300, R)
            ids[v, :len(e)] = e
            deg[v] = len(e)
        a = oracle.inter_insert(X, ids, deg, R)
        b = ref.inter_insert(X, ids, deg, R)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0]), R
        assert a[1].mean() > deg.mean()

