"""In-library hash-sharded index (eps_index_create_sharded, csrc/shard_group.cpp): G per-device indices in one process, row i on
shard i mod G, per-shard top-k pushed peer-to-peer to shard 0's device and merged there.  On the one-GPU test box the shards
share device 0 (`devices=[0, 0, 0]`): every code path is the multi-GPU one except that the peer copies stay on one device."""
import os

import numpy as np
import pytest

from helpers import assert_topk_match, bitset, data

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import vectordb_amd as amd
    from vectordb_amd.build import build
    build()
    return amd


@pytest.mark.parametrize("G", [2, 3])
@pytest.mark.parametrize("metric", [0, 2])
def test_sharded_flat_equals_single_index(amd, oracle, G, metric):
    n, d = 20_011, 48
    X, Q = data(n, d, 3), data(17, d, 4)
    one = amd.GpuIndex(d, metric)
    one.attach_rows(X[:15_000])
    one.append_rows(X[15_000:])
    grp = amd.GpuIndex(d, metric, devices=[0] * G)
    grp.attach_rows(X[:15_000])
    grp.append_rows(X[15_000:15_007])       # appends that do not start on a multiple of G
    grp.append_rows(X[15_007:])
    assert grp.row_count == n
    dele = bitset(n, range(0, n, 7))
    idc = np.arange(n, dtype=np.int32)
    for setup in ("plain", "deleted", "filter"):
        for ix in (one, grp):
            ix.set_deleted(dele if setup == "deleted" else None)
            ix.set_int_filter(idc if setup == "filter" else None, ">=" if setup == "filter" else None, 12_345)
        a = one.search(Q, 10, mode=amd.MODE_FLAT)
        b = grp.search(Q, 10, mode=amd.MODE_FLAT)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), setup
    rid, rd = oracle.topk_flat(metric, X, Q[0], 10)
    one.set_int_filter(None, None, 0)
    grp.set_int_filter(None, None, 0)
    b = grp.search(Q[:1], 10, mode=amd.MODE_FLAT)
    assert_topk_match(b[0][0], b[1][0], rid, rd)
    one.close()
    grp.close()


def test_sharded_large_batch_mfma_engine_and_program_filter(amd):
    """each shard big enough for the MFMA filter engine (>= 65 536 rows), 300 queries, compiled filter over strided attribute rows"""
    n, d, G = 140_000, 64, 2
    X, Q = data(n, d, 5), data(300, d, 6)
    rows = np.zeros(n, dtype=np.dtype([("id", "<i4"), ("price", "<f4")]))
    rows["id"], rows["price"] = np.arange(n), np.random.default_rng(7).random(n)
    prog = [("i32", 0), ("const", 3), ("%",), ("const", 1), ("=",), ("f32", 4), ("const", 0.7), ("<",), ("and",)]
    one = amd.GpuIndex(d, 0)
    grp = amd.GpuIndex(d, 0, devices=[0] * G)
    for ix in (one, grp):
        ix.attach_rows(X)
        ix.set_filter_program(prog, rows)
    a = one.search(Q, 10, mode=amd.MODE_FLAT)
    b = grp.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA)
    assert grp.stats()["rerank_rows"] > 0
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    m = (rows["id"] % 3 == 1) & (rows["price"] < 0.7)
    assert m[a[0]].all()
    one.close()
    grp.close()


def test_sharded_device_buffers_and_rows_attached_per_shard(amd):
    """r4: rows handed over shard by shard where they already live on the shard's device (eps_index_attach_shard_rows: local row l =
    global row l * G + shard), queries and results in device memory (a torch tensor on a device of the group): the per-shard lists are
    merged on the device that holds the caller's result buffers, no host hop, counts included.  Same ids / distances / counts as the
    host-buffer form of the same group and as one plain index; a group whose shards do not hold a hash split of the table refuses."""
    import torch
    n, d, G = 30_001, 40, 3
    X, Q = data(n, d, 31), data(33, d, 32)
    one = amd.GpuIndex(d, 0)
    one.attach_rows(X)
    host = amd.GpuIndex(d, 0, devices=[0] * G)
    host.attach_rows(X)
    grp = amd.GpuIndex(d, 0, devices=[0] * G)
    parts = [torch.from_numpy(np.ascontiguousarray(X[s::G])).cuda() for s in range(G)]
    grp.attach_shard_rows(0, parts[0])
    with pytest.raises(amd.EpsillaError):          # shards 1, 2 still empty: not a hash split of 10 001 rows
        grp.search(Q, 10, mode=amd.MODE_FLAT)
    grp.attach_shard_rows(1, parts[1])
    grp.attach_shard_rows(2, np.ascontiguousarray(X[2::G]))   # host rows of one shard work too
    assert grp.row_count == n
    dele = bitset(n, range(0, n, 5))
    Qd = torch.from_numpy(Q).cuda()
    k = 10
    for setup in ("plain", "deleted"):
        for ix in (one, host, grp):
            ix.set_deleted(dele if setup == "deleted" else None)
        a = one.search(Q, k, mode=amd.MODE_FLAT)
        b = host.search(Q, k, mode=amd.MODE_FLAT)
        o = (torch.full((len(Q), k), -7, dtype=torch.int64, device="cuda"), torch.zeros((len(Q), k), dtype=torch.float32, device="cuda"),
             torch.zeros((len(Q),), dtype=torch.int32, device="cuda"))
        torch.cuda.synchronize()   # (the shards run on their own streams: the fills above must have landed)
        grp.search(Qd, k, out=o, mode=amd.MODE_FLAT)
        grp.synchronize()
        c = (o[0].cpu().numpy(), o[1].cpu().numpy(), o[2].cpu().numpy())
        for got, what in ((b, "host buffers"), (c, "device buffers")):
            assert np.array_equal(a[0], got[0]) and np.array_equal(a[1], got[1]) and np.array_equal(a[2], got[2]), (setup, what)
    # fewer visible rows than k: -1 / +inf padding and counts from the device merge
    few = bitset(n, [i for i in range(n) if i not in (5, 6, 7)])
    grp.set_deleted(few)
    grp.search(Qd[:4], k, out=(o[0][:4], o[1][:4], o[2][:4]), mode=amd.MODE_FLAT)
    grp.synchronize()
    assert o[2][:4].cpu().tolist() == [3] * 4 and sorted(o[0][0, :3].cpu().tolist()) == [5, 6, 7] and (o[0][0, 3:] == -1).all()
    for ix in (one, host, grp):
        ix.close()


def test_sharded_build_graph_search_and_graph_files(amd, tmp_path):
    """one graph per shard (built on the shard's rows), traversal per shard, merged: unique global ids, the same answer after
    a save / load round trip through per-shard files in the reference's ann_graph format"""
    n, d, G = 9_000, 32, 3
    X, Q = data(n, d, 8), data(40, d, 9)
    grp = amd.GpuIndex(d, 0, devices=[0] * G)
    grp.attach_rows(X)
    grp.build(n)
    gn, ge, _ = grp.graph_info()
    assert gn == n and ge > 20 * n
    ids, dist, cnt = grp.search(Q, 10, mode=amd.MODE_GRAPH, intra_threads=4)
    ex = grp.search(Q, 10, mode=amd.MODE_FLAT)
    hits = sum(len(set(ids[i]) & set(ex[0][i])) for i in range(len(Q)))
    assert hits >= 0.98 * 10 * len(Q)
    assert all(len(set(r)) == 10 and r.min() >= 0 and r.max() < n for r in ids) and np.all(np.diff(dist, axis=1) >= 0)
    p = str(tmp_path / "ann_graph_1.bin")
    grp.save_graph(p)
    assert all(os.path.exists(p + ".shard%d" % s) for s in range(G))
    grp2 = amd.GpuIndex(d, 0, devices=[0] * G)
    grp2.attach_rows(X)
    grp2.load_graph(p)
    ids2, dist2, _ = grp2.search(Q, 10, mode=amd.MODE_GRAPH, intra_threads=4)
    assert np.array_equal(ids, ids2) and np.array_equal(dist, dist2)
    with pytest.raises(amd.EpsillaError):
        grp.set_graph(np.zeros(2, np.int64), np.zeros(1, np.int64), 0)
    grp.close()
    grp2.close()


def test_dropin_dbserver_over_a_sharded_executor(tmp_path):
    """EPS_DEVICES=0,0: the reference DBServer on the drop-in executor with its table hash-sharded over two shards - exact
    flat answers, device-compiled and host-evaluated filters, deletes; equal to the reference DBServer's own (exact at this size)."""
    from oracle.pyoracle import DROPIN_SO, Ref, dropin_available, ref_available
    if not (dropin_available() and ref_available()):
        pytest.skip("needs dropin/_build and oracle/_ref")
    os.environ["EPS_DEVICES"] = "0,0"
    try:
        ref, drop = Ref(), Ref(DROPIN_SO)
        schema = {"name": "T", "fields": [{"name": "ID", "dataType": "INT", "primaryKey": True}, {"name": "Tag", "dataType": "STRING"},
                                           {"name": "Price", "dataType": "FLOAT"},
                                           {"name": "V", "dataType": "VECTOR_FLOAT", "dimensions": 12, "metricType": "EUCLIDEAN"}]}
        n = 1501
        X = data(n, 12, 11)
        price = np.random.default_rng(12).random(n)
        recs = [{"ID": int(i), "Tag": "t%d" % (i % 4), "Price": float(np.float32(price[i])), "V": [float(x) for x in X[i]]} for i in range(n)]
        Q = data(5, 12, 13)
        outs = []
        for lib, name in ((ref, "ref"), (drop, "drop")):
            lib.L.ref_config(1, 500, 1, 0, 2)
            db = lib.db(str(tmp_path / name))
            assert db.create_table(schema) == 0 and db.insert("T", recs[:900]) == 0 and db.insert("T", recs[900:]) == 0
            assert db.delete("T", [0, 1, 2, 700]) == 0
            res = {}
            for flt in ("", "ID >= 400 AND Price < 0.5", "Tag = 't2'", "Tag = 't1' OR ID < 100", "@distance < 0.9"):
                for qi, q in enumerate(Q):
                    res[(flt, qi)] = db.search("T", "V", q, 20, fields=("ID",), flt=flt)
            outs.append(res)
            db.close()
            lib.L.ref_config(4, 500, 1, 0, 4)
        for key, (rc, r) in outs[0].items():
            rc2, d = outs[1][key]
            assert rc == rc2 == 0, (key, r if rc else d)
            assert [x["ID"] for x in d] == [x["ID"] for x in r], key
            assert np.allclose([x["@distance"] for x in d], [x["@distance"] for x in r], rtol=1e-4)
    finally:
        os.environ.pop("EPS_DEVICES", None)


def test_dropin_rebuild_builds_per_shard_graphs_on_the_mirror(tmp_path):
    """EPS_DEVICES=0,0 + Rebuild(): ANNGraphSegment::BuildFromVectorTable builds on the field's device mirror - every shard the
    graph of its own rows, on its own device - and the executors then walk the per-shard graphs (SearchQueueSize 500 over 1250-row
    shards evaluates nearly everything: recall >= 0.99 against the reference DBServer's exact answers); rows inserted after the
    rebuild are found through the shards' brute-force tails."""
    from oracle.pyoracle import DROPIN_SO, Ref, dropin_available, ref_available
    if not (dropin_available() and ref_available()):
        pytest.skip("needs dropin/_build and oracle/_ref")
    os.environ["EPS_DEVICES"] = "0,0"
    try:
        ref, drop = Ref(), Ref(DROPIN_SO)
        schema = {"name": "T", "fields": [{"name": "ID", "dataType": "INT", "primaryKey": True},
                                           {"name": "V", "dataType": "VECTOR_FLOAT", "dimensions": 16, "metricType": "EUCLIDEAN"}]}
        n = 2500
        X = data(n + 40, 16, 21)
        recs = [{"ID": int(i), "V": [float(x) for x in X[i]]} for i in range(n + 40)]
        Q = data(24, 16, 22)
        outs = []
        for lib, name in ((ref, "ref"), (drop, "drop")):
            db = lib.db(str(tmp_path / name))
            assert db.create_table(schema) == 0 and db.insert("T", recs[:n]) == 0
            if lib is drop:
                assert db.rebuild() == 0
            assert db.insert("T", recs[n:]) == 0
            outs.append([db.search("T", "V", q, 10, fields=("ID",)) for q in Q])
            db.close()
        hits = 0
        for (rc, r), (rc2, d) in zip(*outs):
            assert rc == rc2 == 0
            hits += len(set(x["ID"] for x in r) & set(x["ID"] for x in d))
        assert hits >= 0.99 * 10 * len(Q), hits
        assert any(x["ID"] >= n for rc, d in outs[1] for x in d) == any(x["ID"] >= n for rc, r in outs[0] for x in r)
    finally:
        os.environ.pop("EPS_DEVICES", None)


def test_library_owned_exchange_on_rccl_world_of_one(amd):
    """eps_exchange (include/epsilla_gfx950.h, r6): the library's own RCCL communicator, ncclAllGather of the packed lists, k-way merge.  One GPU on
    this box, so the communicator has ONE rank (RCCL refuses two ranks on a device): that still runs the whole chain - dlopen of RCCL (the copy the
    process already holds: torch's), unique id, ncclCommInitRank, the in-place / out-of-place all-gather, merge_shards_kernel, the event ring - and
    the merged answer of a world of one is its own lists.  Several batch shapes, packed and separate buffers, -1 padded short lists."""
    import torch
    dev = torch.device("cuda", 0)
    x = amd.Exchange(0, 1, amd.Exchange.unique_id(), device=0)
    info = x.info()
    assert info["world"] == 1 and info["rccl_version"] > 0 and "rccl" in info["rccl_library"], info
    g = torch.Generator(device=dev).manual_seed(3)
    for nq, k in ((1, 10), (7, 3), (1024, 10), (33, 100)):
        dist = torch.sort(torch.rand((nq, k), generator=g, device=dev), dim=1).values
        ids = torch.randint(0, 1 << 40, (nq, k), generator=g, device=dev, dtype=torch.int64)
        if nq > 1:
            ids[1, k // 2:] = -1                       # a short list: -1 beyond its count
        pack = torch.empty(((nq * k * 12 + 7) // 8 * 8,), dtype=torch.uint8, device=dev)   # the packed layout bench.py searches into
        p_ids = pack[: nq * k * 8].view(torch.int64).view(nq, k)
        p_dd = pack[nq * k * 8: nq * k * 12].view(torch.float32).view(nq, k)
        p_ids.copy_(ids)
        p_dd.copy_(dist)
        for a_ids, a_dd in ((ids, dist), (p_ids, p_dd)):
            o_i = torch.full((nq, k), 7, dtype=torch.int64, device=dev)
            o_d = torch.zeros((nq, k), dtype=torch.float32, device=dev)
            x.allgather_merge(a_ids, a_dd, o_i, o_d, stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            want_i, want_d = ids.clone(), dist.clone()
            if nq > 1:
                want_d[1, k // 2:] = float("inf")
            assert torch.equal(o_i, want_i) and torch.equal(o_d, want_d), (nq, k)
    t = x.times_us(8)
    assert len(t) == 8 and all(a >= 0 and b >= 0 for a, b in t)
    x.close()


@pytest.mark.parametrize("world", [2, 4])
def test_mailbox_exchange_between_processes(amd, world):
    """"for b=1 prefer direct P2P stores into a peer-mapped buffer + flag" - between PROCESSES (r6; the one-process shard group had it
    since r4).  `world` processes on this box's one GPU: every rank exports its mailbox (hipIpc), maps its peers', and every call stores its packed
    lists into every peer's mailbox, raises a flag, waits on the device for all flags of its own mailbox and merges (eps_exchange_direct_merge) - no
    collective, no host round trip.  60 calls of 1..100 queries, k 10..100, some left in flight so that consecutive calls overlap across ranks (the
    two slot parities); every rank checks every merged answer against numpy.  (tests/workers/mailbox_rank.py)"""
    import subprocess
    import sys
    port = 29610 + world
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "workers", "mailbox_rank.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(port), "60"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
             for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("rank %d ok" % r) in o, (r, p.returncode, o[-1500:])
