"""Seeded random configurations: whatever engine FLAT_AUTO picks (stream scan, MFMA filter v7<1>/v7<2>/v3, in slices or
not, seeded or not, with or without a deleted bitset / attribute filter) must return the stream engine's bits."""
import os

import numpy as np
import pytest

from helpers import assert_topk_match, data


@pytest.fixture(scope="module")
def amd():
    import vectordb_amd
    from vectordb_amd.build import build
    build()
    return vectordb_amd


def _configs():
    rng = np.random.default_rng(20240924 + int(os.environ.get("EPS_FUZZ_SEED", "0")))   # (EPS_FUZZ_SEED: other shapes for a longer soak)
    dims = [64, 96, 128, 200, 256, 320, 384, 512, 640, 1000]
    out = []
    for i in range(18):
        out.append(dict(n=int(rng.integers(70_000, 260_000)), d=int(rng.choice(dims)), nq=int(rng.choice([1, 5, 9, 40, 129, 300, 600, 2100])),
                        k=int(rng.choice([1, 5, 10, 37, 100, 128])), metric=int(rng.integers(0, 3)),
                        flt=str(rng.choice(["none", "none", "deleted50", "deleted90", "int"])), seed=1000 + i))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("c", _configs(), ids=lambda c: "n%d-d%d-q%d-k%d-m%d-%s" % (c["n"], c["d"], c["nq"], c["k"], c["metric"], c["flt"]))
def test_auto_engine_equals_stream_engine(amd, c):
    rng = np.random.default_rng(c["seed"])
    X, Q = data(c["n"], c["d"], c["seed"]), data(c["nq"], c["d"], c["seed"] + 1)
    if c["metric"] == 1:
        X = amd.normalize_rows(X, only_if_nonzero=True)
        Q = amd.normalize_rows(Q, only_if_nonzero=False)
    ix = amd.GpuIndex(c["d"], c["metric"])
    ix.attach_rows(X)
    vis = np.ones(c["n"], bool)
    if c["flt"].startswith("deleted"):
        dele = rng.random(c["n"]) < (0.5 if c["flt"] == "deleted50" else 0.9)
        ix.set_deleted(np.packbits(dele, bitorder="little"))
        vis &= ~dele
    elif c["flt"] == "int":
        col = rng.integers(-1000, 1000, c["n"]).astype(np.int32)
        ix.set_int_filter(col, "<", 250)
        vis &= col < 250
    a = ix.search(Q, c["k"], mode=amd.MODE_FLAT, flat_engine=amd.FLAT_AUTO)
    b = ix.search(Q, c["k"], mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
    m = ix.search(Q, c["k"], mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA)
    m8 = ix.search(Q, c["k"], mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
    for got in (a, m, m8):
        assert np.array_equal(got[0], b[0]), "%d result ids differ" % (got[0] != b[0]).sum()
        assert np.array_equal(got[1], b[1]) and np.array_equal(got[2], b[2])
    ids = b[0][b[0] >= 0]
    assert vis[ids].all() and (b[2] == min(c["k"], int(vis.sum()))).all()
    ix.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(24)))
def test_fuzz_traversal_matches_oracle_lockstep(amd, oracle, monkeypatch, seed):
    """(Odd seeds run with the 8-bit prefilter forced on, even seeds with it off.)
    Seeded fuzz of the traversal kernel over worker counts, queue sizes (LocalQueueSize != SearchQueueSize included), sync
    intervals, metrics, batch sizes (4- and 16-wavefront variants) and adjacency shapes (fixed stride; CSR with lists beyond 64
    entries, duplicates and empty lists) against the oracle's SearchImpl under the lockstep schedule: whole returned prefix of the
    master queue and the evaluation count."""
    monkeypatch.setenv("EPS_TRV_PREFILTER", str(seed & 1))
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(600, 1500))
    d = int(rng.choice([8, 24, 33, 64]))
    metric = int(rng.integers(0, 3))
    X = rng.random((n, d), dtype=np.float32)
    nq = int(rng.choice([1, 5, 300]))
    Q = rng.random((nq, d), dtype=np.float32)
    if metric == 1:
        X /= np.linalg.norm(X, axis=1, keepdims=True)
        Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    off, nbr, nav = oracle.build_graph(0, X, K=int(rng.integers(20, 60)), out_degree=int(rng.integers(10, 40)))
    lists = [list(nbr[off[i]:off[i + 1]]) for i in range(n)]
    shape = int(rng.integers(0, 3))
    if shape >= 1:                                  # a few long lists (CSR form on the device), duplicates, empty lists
        for v in rng.choice(n, size=6, replace=False):
            lists[int(v)] += [int(x) for x in rng.integers(0, n, size=int(rng.integers(70, 200)))]
        for v in rng.choice(n, size=10, replace=False):
            if lists[int(v)]:
                lists[int(v)].append(lists[int(v)][0])
    if shape == 2:
        for v in rng.choice(n, size=5, replace=False):
            if int(v) != nav:
                lists[int(v)] = []
    off2 = np.zeros(n + 1, np.int64)
    off2[1:] = np.cumsum([len(l) for l in lists])
    nbr2 = np.asarray([x for l in lists for x in l], np.int64)
    T = int(rng.choice([1, 2, 3, 4, 8]))
    L = int(rng.choice([40, 128, 500, 800]))
    L = min(L, n)
    Lq = int(rng.choice([L, max(8, L // 3)]))
    I = int(rng.choice([1, 4, 15]))
    k = int(min(Lq, rng.choice([10, 64])))
    ix = amd.GpuIndex(d, metric)
    ix.attach_rows(X)
    ix.set_graph(off2, nbr2, nav)
    ids, dist, cnt = ix.search(Q, k, mode=amd.MODE_GRAPH, intra_threads=T, master_queue=L, local_queue=Lq, sync_interval=I)
    ev_gpu = ix.stats()["dist_evals"]
    init = oracle.prepare_init_ids(off2, nbr2, nav, L)
    ev_or = 0
    step = max(1, nq // 40)
    for qi in range(nq):
        if qi % step and nq > 40:
            continue
        oid, od, ev = oracle.search_impl(metric, X, off2, nbr2, init, Q[qi], T=T, L=L, Lq=Lq, I=I, lockstep=True)
        ev_or += ev
        assert int(cnt[qi]) == k, (seed, qi)
        assert_topk_match(ids[qi], dist[qi], oid[:k], od[:k], what="fuzz %d: n%d d%d m%d T%d L%d Lq%d I%d shape%d q%d" % (seed, n, d, metric, T, L, Lq, I, shape, qi))
    if nq <= 40:
        assert abs(ev_gpu - ev_or) <= max(2, ev_or // 100), (seed, ev_gpu, ev_or)
    ix.close()
