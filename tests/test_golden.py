"""Oracle vs. committed golden vectors (tests/golden/*.npz, produced from the compiled reference by
scripts/gen_golden.py) and vs. the known-answer cases of the reference's own gtest suite
(engine/test/engine/db/db_server.cpp).  CPU only; does not need /root/reference."""
import os

import numpy as np

from oracle.pyoracle import make_filter

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def data(n, d, seed):
    return np.random.default_rng(seed).random((n, d), dtype=np.float32)


def test_distances_golden(oracle):
    z = np.load(os.path.join(G, "distances.npz"))
    for d in (1, 3, 4, 7, 33, 128, 768):
        X, Q = data(32, d, 100 + d), data(3, d, 200 + d) * 2 - 0.5
        for m in (0, 1, 2):
            got = np.stack([oracle.dist_batch(m, X, q) for q in Q])
            assert np.array_equal(got, z["d%d_m%d" % (d, m)])


def test_search_impl_golden(oracle):
    z = np.load(os.path.join(G, "graph2000x32.npz"))
    X, Q = data(2000, 32, 42), data(16, 32, 43)
    off, nbr, nav = z["off"].astype(np.int64), z["nbr"].astype(np.int64), int(z["nav"])
    init = oracle.prepare_init_ids(off, nbr, nav, 500)
    assert np.array_equal(init, z["init"])
    for m in (0, 1, 2):
        for qi, q in enumerate(Q):
            ids, ds, _ = oracle.search_impl(m, X, off, nbr, init, q, T=1, L=500)
            assert np.array_equal(ids, z["ids_m%d" % m][qi])
            assert np.array_equal(ds, z["dist_m%d" % m][qi])


def test_nsg_golden(oracle):
    z = np.load(os.path.join(G, "nsg600x16.npz"))
    X = data(600, 16, 7)
    off, nbr, nav = oracle.nsg_build(X, oracle.knn_exact(0, X, 100), seed=100)
    assert nav == int(z["nav"]) and np.array_equal(off, z["off"]) and np.array_equal(nbr, z["nbr"])


def test_search_graph_tail_filter_deleted_golden(oracle):
    """Full Search(): graph over rows [0,1000) + brute-force tail [1000,1500) + post-filter + deletes."""
    z = np.load(os.path.join(G, "dbserver1500x8.npz"))
    X, Q = data(1500, 8, 15), data(8, 8, 16)
    off, nbr, nav = z["off"].astype(np.int64), z["nbr"].astype(np.int64), int(z["nav"])
    idc = np.arange(1500, dtype=np.int32)
    cases = {"plain": make_filter(), "lt700": make_filter(attr=idc, op="<", value=700),
             "ge1200": make_filter(attr=idc, op=">=", value=1200)}
    for name, (flt, keep) in cases.items():
        for limit in (10, 100):
            for qi, q in enumerate(Q):
                ids, ds, _ = oracle.search(0, X, 1000, off, nbr, nav, q, limit, T=1, L=500, flt=flt)
                want = z["%s_k%d_ids" % (name, limit)][qi]
                m = int((want >= 0).sum())
                assert len(ids) == m and np.array_equal(ids, want[:m])
                assert np.array_equal(ds, z["%s_k%d_dist" % (name, limit)][qi][:m])
    bits = np.zeros((1500 + 7) // 8, np.uint8)
    for i in range(0, 1500, 3):
        bits[i >> 3] |= 1 << (i & 7)
    flt, keep = make_filter(deleted=bits)
    for qi, q in enumerate(Q):
        ids, ds, _ = oracle.search(0, X, 1000, off, nbr, nav, q, 10, T=1, L=500, flt=flt)
        want = z["deleted3_k10_ids"][qi]
        m = int((want >= 0).sum())
        assert np.array_equal(ids, want[:m]) and np.array_equal(ds, z["deleted3_k10_dist"][qi][:m])


# ---- the reference's own known-answer tests --------------------------------------------------------------
CITIES = [("Berlin", [0.05, 0.61, 0.76, 0.74]), ("London", [0.19, 0.81, 0.75, 0.11]),
          ("Moscow", [0.36, 0.55, 0.47, 0.94]), ("San Francisco", [0.18, 0.01, 0.85, 0.80]),
          ("Shanghai", [0.24, 0.18, 0.22, 0.44])]


def test_known_answer_dense_vector(oracle):
    """DbServer.DenseVector (db_server.cpp:92-319): expected full orderings at :289-292."""
    X = np.array([v for _, v in CITIES], np.float32)
    q = np.array([0.35, 0.55, 0.47, 0.94], np.float32)
    want = {0: ["Moscow", "Berlin", "Shanghai", "San Francisco", "London"],
            2: ["Moscow", "Berlin", "San Francisco", "London", "Shanghai"],
            1: ["Moscow", "Shanghai", "Berlin", "San Francisco", "London"]}
    for metric, order in want.items():
        rows, qq = X, q
        if metric == 1:
            rows = np.stack([oracle.normalize_insert(x) for x in X])
            qq = oracle.normalize_query(q)
        ids, ds, _ = oracle.search(metric, rows, 0, None, None, 0, qq, 6)
        assert [CITIES[i][0] for i in ids] == order
    # survey probe anchors: squared L2, negated dot, 1-dot on normalised rows
    ids, ds, _ = oracle.search(0, X, 0, None, None, 0, q, 1)
    assert abs(ds[0] - 1.0000040e-4) < 1e-9
    ids, ds, _ = oracle.search(2, X, 0, None, None, 0, q, 1)
    assert abs(ds[0] + 1.5329999924) < 1e-6


def test_known_answer_delete_by_pk(oracle):
    """DbServer.DeleteByPK (db_server.cpp:514-751): after deleting PKs 1-4 only Shanghai is returned."""
    X = np.array([v for _, v in CITIES], np.float32)
    q = np.array([0.35, 0.55, 0.47, 0.94], np.float32)
    flt, keep = make_filter(deleted=np.array([0b1111], np.uint8))
    ids, ds, _ = oracle.search(0, X, 0, None, None, 0, q, 6, flt=flt)
    assert [CITIES[i][0] for i in ids] == ["Shanghai"]
    assert abs(ds[0] - 0.46149999) < 1e-6


def test_known_answer_filter(oracle):
    """DbServer.DenseVectorFilter (db_server.cpp:1407-1630): `ID <= 2` returns exactly 2 rows."""
    X = np.array([v for _, v in CITIES], np.float32)
    q = np.array([0.35, 0.55, 0.47, 0.94], np.float32)
    flt, keep = make_filter(attr=np.arange(1, 6, dtype=np.int32), op="<=", value=2)
    ids, ds, _ = oracle.search(0, X, 0, None, None, 0, q, 6, flt=flt)
    assert sorted(int(i) + 1 for i in ids) == [1, 2]


def test_known_answer_unit_circle(oracle):
    """DbServer.QueryDenseVectorDuringRebuild (db_server.cpp:1085-1245), scaled to N=2000: unit vectors
    (cos(pi i/N), sin(pi i/N)), COSINE, query (1,0), limit 500 -> ids 0..499 in order, through the graph
    built on the first half plus the brute-force tail."""
    N = 2000
    i = np.arange(N)
    X = np.stack([np.cos(np.pi * i / N), np.sin(np.pi * i / N)], 1).astype(np.float32)
    X = np.stack([oracle.normalize_insert(x) for x in X])
    q = oracle.normalize_query(np.array([1.0, 0.0], np.float32))
    half = N // 2
    off, nbr, nav = oracle.build_graph(1, X[:half])
    ids, ds, _ = oracle.search(1, X, half, off, nbr, nav, q, 500, T=1, L=500)
    assert np.array_equal(ids, np.arange(500))
    ids, ds, _ = oracle.search(1, X[:half], half, off, nbr, nav, q, 500, T=4, L=500)
    assert np.array_equal(ids, np.arange(500))
