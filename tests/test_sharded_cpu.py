"""N > 1 path on CPU: world_size-2 gloo.  The device kernels cannot run here, so each rank produces its shard's
exact top-k with the CPU oracle (checker standing in for eps_index_search), and the REAL sharding logic is
exercised: hash-sharding by row index, global id = local*G + rank, one all-gather of [nq,k] (dist,id), k-way merge
by (dist,id) (a numpy restatement of eps_merge_topk, whose device version is checked in test_gpu_parity).  The
merged answer must equal the unsharded exact answer.

The product code of the N > 1 path itself (eps_index_set_id_map, the packed all-gather buffer, eps_merge_topk_packed, the
in-library shard group) cannot run without a GPU; it is covered on the GPU box with product code on both sides by
tests/test_bench_contract.py::test_bench_two_ranks_on_one_gpu_merge_equals_unsharded_scan (two gloo ranks on one GPU through
bench.py) and tests/test_gpu_sharded.py (eps_index_create_sharded with several shards on one device)."""
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def merge_topk_np(dist, ids, k):
    """dist/ids: [shards, nq, k] -> merged [nq, k] by (dist, id); id -1 = empty."""
    S, nq, _ = dist.shape
    out_d = np.full((nq, k), np.inf, np.float32)
    out_i = np.full((nq, k), -1, np.int64)
    for q in range(nq):
        pairs = sorted((float(dist[s, q, e]), int(ids[s, q, e])) for s in range(S) for e in range(dist.shape[2]) if ids[s, q, e] >= 0)
        for e, (d, i) in enumerate(pairs[:k]):
            out_d[q, e], out_i[q, e] = d, i
    return out_d, out_i


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle.pyoracle import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = Oracle()
    n, d, nq, k = 3001, 24, 9, 10
    X = np.random.default_rng(42).random((n, d), dtype=np.float32)
    Q = np.random.default_rng(43).random((nq, d), dtype=np.float32)
    shard = X[rank::world]                         # row i lives on rank i mod G
    loc_d = np.full((nq, k), np.inf, np.float32)
    loc_i = np.full((nq, k), -1, np.int64)
    for qi in range(nq):
        ids, ds = orc.topk_flat(0, shard, Q[qi], k)
        loc_i[qi, :len(ids)] = ids * world + rank  # eps_index_set_id_map(rank, world)
        loc_d[qi, :len(ids)] = ds
    g_d = [torch.empty((nq, k), dtype=torch.float32) for _ in range(world)]
    g_i = [torch.empty((nq, k), dtype=torch.int64) for _ in range(world)]
    dist.all_gather(g_d, torch.from_numpy(loc_d))
    dist.all_gather(g_i, torch.from_numpy(loc_i))
    md, mi = merge_topk_np(np.stack([t.numpy() for t in g_d]), np.stack([t.numpy() for t in g_i]), k)
    ok = True
    for qi in range(nq):
        ids, ds = orc.topk_flat(0, X, Q[qi], k)
        ok &= bool(np.array_equal(mi[qi], ids) and np.array_equal(md[qi], ds))
    t = torch.tensor([1 if ok else 0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        q.put(int(t.item()))
    dist.destroy_process_group()


def test_hash_sharded_search_merges_to_the_unsharded_answer():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) == 1


def test_merge_restatement_handles_short_lists():
    d = np.array([[[0.1, 0.5, np.inf]], [[0.2, 0.3, 0.4]]], np.float32)
    i = np.array([[[7, 9, -1]], [[2, 4, 6]]], np.int64)
    md, mi = merge_topk_np(d, i, 4)
    assert list(mi[0]) == [7, 2, 4, 6] and np.allclose(md[0], [0.1, 0.2, 0.3, 0.4])
