#!/usr/bin/env python
"""Headline benchmark: QPS at recall@10 >= 0.999 on 10M x 768 L2, k=10, batch=1024 (BASELINE.json configs[2]),
synthetic i.i.d. U[0,1) fp32 rows generated on the device, inputs resident in HBM when the timed region starts.

    python bench.py [--gpus N --steps K --warmup W]       (N > 1 is launched by torch.distributed.run)

One "step" = one batch of queries through the hot path (eps_index_search: flat scan or graph traversal -> top-k).
With N > 1 the corpus is hash-sharded by row index (row i lives on rank i mod N), every rank answers the same query
batch on its shard, and the per-shard top-k lists are merged after ONE RCCL all-gather of [batch,k] (dist, id)
pairs.  Weak scaling, per-GPU work fixed, in one of two forms (--scale):
  queries (default): the corpus stays --rows (10M) in total, each rank holds rows/N of it, and the batch grows to
                     N x --batch - the whole-job queries/s (`value`) grows with N;
  rows:              every rank holds --rows rows (80M rows at N = 8) and the batch stays --batch - queries/s
                     stays flat while the corpus grows; `work_rate` (query*rows/s) is the number that scales.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# HBM bytes per launch of the dominant kernel from the PMC counters of profiles/r1_pmc_10Mx768_b1024.csv, collected
# in separate --pmc passes and corrected as MI355X_MICROARCH.md prescribes (gfx950 FETCH_SIZE counts 128-B requests at
# 64 B: x2; FETCH_SIZE/WRITE_SIZE are in KiB): (2 * 7042014 + 11837) KiB for the 9,257,600-row launch of
# mfma_filter_kernel_v7 = 14.43e9 bytes, against 14.22e9 algorithmic bytes of the fp16 mirror (each row tile is fetched
# from HBM once; without the per-tile rendezvous of the workgroups that share a row tile it was 27.4e9).
TRAFFIC = {"mfma": (2 * 7042014 + 11837) * 1024.0}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_F16_PEAK_TF = 2500.0  # dense bf16/f16 MFMA peak (nominal, 2.4 GHz)
MFMA_F16_SUSTAINED_TF = 1814.0  # measured: v_mfma_f32_32x32x16_f16 alone, operands toggling like data, 1.82 GHz (scripts/lab/mfma_peak.hip)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", type=int, default=10_000_000, help="rows per GPU")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--metric", default="EUCLIDEAN")
    ap.add_argument("--mode", default="flat", choices=["flat", "graph"])
    ap.add_argument("--engine", default="auto", choices=["auto", "stream", "mfma"])
    ap.add_argument("--recall-queries", type=int, default=32)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--scale", default="queries", choices=["queries", "rows"],
                    help="N > 1: 'queries' = the --rows corpus is hash-sharded over the N GPUs and the batch grows to "
                         "N x --batch (per-GPU work fixed, whole-job queries/s grows with N); 'rows' = every GPU holds "
                         "--rows rows (corpus grows to N x --rows, SURVEY C5) and the batch stays --batch")
    return ap.parse_args()


def gen_rows(torch, n, d, seed, device):
    """i.i.d. U[0,1) fp32, generated on the device in slabs (seeded per rank)."""
    g = torch.Generator(device=device).manual_seed(seed)
    X = torch.empty((n, d), dtype=torch.float32, device=device)
    step = 1 << 20
    for s in range(0, n, step):
        e = min(n, s + step)
        X[s:e] = torch.rand((e - s, d), generator=g, device=device, dtype=torch.float32)
    return X


def exact_topk_torch(torch, X, q, k, id_base, id_stride):
    """fp32 direct-form exact scan of ONE query in torch (ground truth for recall; not timed)."""
    best_d, best_i = None, None
    step = 1 << 20
    for s in range(0, X.shape[0], step):
        e = min(X.shape[0], s + step)
        dd = ((X[s:e] - q) ** 2).sum(1)
        kk = min(k, e - s)
        v, i = torch.topk(dd, kk, largest=False)
        i = (i + s) * id_stride + id_base
        if best_d is None:
            best_d, best_i = v, i
        else:
            v = torch.cat([best_d, v])
            i = torch.cat([best_i, i])
            o = torch.argsort(v, stable=True)[:k]
            best_d, best_i = v[o], i[o]
    return best_d, best_i


def cpu_baseline(args, budget_s):
    """The reference's own CPU distance path (oracle/_ref = reference sources compiled verbatim; falls back to the
    plain-C oracle port if that build is absent) timed on this box's host cores on a bounded sample of the same
    workload: full flat scans (distance over all cores + top-k) of a row sample, scaled to the configured row
    count.  A reported baseline only — never part of the product path."""
    from oracle import pyoracle
    cores = os.cpu_count() or 1
    d, k = args.dim, args.k
    rng = np.random.default_rng(42)
    sample_rows = min(args.rows, 400_000)
    X = rng.random((sample_rows, d), dtype=np.float32)
    Q = rng.random((64, d), dtype=np.float32)
    if pyoracle.ref_available():
        ref = pyoracle.Ref()
        kind, threads = "reference", int(ref.L.ref_omp_max_threads())
        scan = lambda q: ref.dist_batch(0, X, q)  # GetDistFunc(L2Sqr) under `omp parallel for`, as BruteForceSearch :729-735
    else:
        orc = pyoracle.Oracle()
        kind, threads = "port", 1
        scan = lambda q: orc.dist_batch(0, X, q)
    scan(Q[0])
    t0 = time.time()
    done = 0
    while done < len(Q) and time.time() - t0 < budget_s:
        dist = scan(Q[done])
        idx = np.argpartition(dist, k)[:k]
        idx[np.argsort(dist[idx], kind="stable")]
        done += 1
    sec = time.time() - t0
    rows_per_s = done * sample_rows / sec
    qps = rows_per_s / args.rows
    return {"value": qps, "unit": "queries/s", "cores": threads, "kind": kind,
            "sample": "%d full flat scans (reference fvec_L2sqr via GetDistFunc, omp over %d threads, + top-%d) of a "
                      "%d x %d row sample in %.1f s, scaled linearly to %d rows; host has %d logical cores"
                      % (done, threads, k, sample_rows, d, sec, args.rows, cores)}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import vectordb_amd as amd
    from vectordb_amd.build import build

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # EPS_BENCH_BACKEND=gloo lets the N > 1 path be exercised with several ranks on ONE GPU (exchange staged through the
    # host); the real multi-GPU run uses nccl (= RCCL over xGMI)
    backend = os.environ.get("EPS_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    def all_gather(dst, src):
        if backend == "nccl":
            dist.all_gather_into_tensor(dst, src)
        else:
            hs = src.cpu()
            parts = [torch.empty_like(hs) for _ in range(world)]
            dist.all_gather(parts, hs)
            dst.copy_(torch.stack(parts).reshape(dst.shape))
    if rank == 0:
        build()
    if world > 1:
        dist.barrier()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    n, d, b, k = args.rows, args.dim, args.batch, args.k
    if world > 1 and args.scale == "queries":
        n = args.rows // world          # this rank's shard of the fixed corpus (row i lives on rank i mod world)
        b = args.batch * world          # every rank answers the whole (larger) batch on its shard

    X = gen_rows(torch, n, d, 42 + rank, dev)                  # this rank's shard: global row id = local*world + rank
    gq = torch.Generator(device=dev).manual_seed(43)           # same queries on every rank
    queries = [torch.rand((b, d), generator=gq, device=dev, dtype=torch.float32) for _ in range(args.steps + args.warmup)]

    ix = amd.GpuIndex(d, args.metric, device=local_rank)
    ix.set_stream(torch.cuda.current_stream().cuda_stream)
    ix.attach_rows(X)
    ix.set_id_map(rank, world)
    engine = {"auto": amd.FLAT_AUTO, "stream": amd.FLAT_STREAM, "mfma": amd.FLAT_MFMA}[args.engine]
    mode = amd.MODE_FLAT if args.mode == "flat" else amd.MODE_GRAPH
    if args.mode == "graph":
        ix.build(n)

    ids = torch.empty((b, k), dtype=torch.int64, device=dev)
    dd = torch.empty((b, k), dtype=torch.float32, device=dev)
    cnt = torch.empty((b,), dtype=torch.int32, device=dev)
    if world > 1:
        g_d = torch.empty((world, b, k), dtype=torch.float32, device=dev)
        g_i = torch.empty((world, b, k), dtype=torch.int64, device=dev)
        m_d = torch.empty((b, k), dtype=torch.float32, device=dev)
        m_i = torch.empty((b, k), dtype=torch.int64, device=dev)

    main_ms, launches, kq = [], [], []

    def step(q):
        ix.search(q, k, out=(ids, dd, cnt), mode=mode, flat_engine=engine)
        if world > 1:
            # the one exchange step of the path: all-gather of the per-shard top-k, then a k-way merge
            all_gather(g_d, dd)
            all_gather(g_i, ids)
            amd.merge_topk(g_d, g_i, m_d, m_i, device=local_rank, stream=torch.cuda.current_stream().cuda_stream)
            return m_d, m_i
        return dd, ids

    for w in range(args.warmup):
        step(queries[w])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        out_d, out_i = step(queries[args.warmup + s])
        st = ix.stats()  # reads the hipEvent pair the library recorded around its dominant kernel on this stream
        main_ms.append(st["main_kernel_ms"])
        launches.append(st["main_kernel_rows"])
        kq.append(st.get("main_kernel_queries", b) or b)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    st = ix.stats()

    # recall@10 of the last batch against an exact fp32 scan (untimed)
    nrec = min(args.recall_queries, b)
    hits = 0
    qlast = queries[-1]
    for qi in range(nrec):
        gd, gi = exact_topk_torch(torch, X, qlast[qi], k, rank, world)
        if world > 1:
            ad = torch.empty((world, k), dtype=torch.float32, device=dev)
            ai = torch.empty((world, k), dtype=torch.int64, device=dev)
            all_gather(ad, gd.contiguous())
            all_gather(ai, gi.contiguous())
            o = torch.argsort(ad.flatten(), stable=True)[:k]
            gi = ai.flatten()[o]
        hits += len(set(gi.tolist()) & set(out_i[qi].tolist()))
    recall = hits / float(nrec * k)

    if rank == 0:
        qps = b * args.steps / elapsed
        kernel_ms = float(np.mean(main_ms)) if main_ms else 0.0   # hipEvent pair around the dominant launch
        krows = float(np.mean(launches)) if launches else 0.0     # rows that launch covered
        used_mfma = st.get("rerank_rows", 0) > 0
        if used_mfma:
            # algorithmic flops of the timed launch: 2 * batch * rows * d, on the fp16 dense MFMA roof
            flops = 2.0 * float(np.mean(kq)) * krows * d   # queries x rows of the timed launch (batches > 2048 run in slices)
            roof = {"bound": "mfma", "kernel": "mfma_filter_kernel_v7 (largest of the 3 filter stages: %d of %d rows x %d of %d queries)" % (krows, n, int(np.mean(kq)), b),
                    "achieved": flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms else None,
                    "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s",
                    "traffic": TRAFFIC.get("mfma") if (world == 1 and n == 10_000_000 and b == 1024 and d == 768) else None}
        elif args.mode == "graph":
            # E evaluations (a row + its id) and X expansions (offsets + an adjacency list) per launch
            alg_bytes = st["dist_evals"] * (4.0 * d + 4) + st["expansions"] * (8 + 4 * 50.0)
            roof = {"bound": "hbm", "kernel": "traverse_kernel (gather: E*(4d+4) + X*(8+4*deg) bytes)",
                    "achieved": alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms else None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None}
        else:
            # a flat scan needs rows*4*d bytes ONCE per batch; the stream engine re-reads the store once
            # per group of 4 queries, which this figure deliberately does not credit.
            alg_bytes = krows * 4 * d
            roof = {"bound": "hbm", "kernel": "flat_scan_kernel", "achieved": alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms else None,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None}
        roof["frac"] = (roof["achieved"] / roof["peak"]) if roof["achieved"] else None
        if used_mfma and roof["achieved"]:
            roof["sustained_peak_measured"] = MFMA_F16_SUSTAINED_TF
            roof["frac_of_sustained"] = roof["achieved"] / MFMA_F16_SUSTAINED_TF
        roof["kernel_ms_per_step"] = kernel_ms * (b / float(np.mean(kq)) if used_mfma and kq else 1.0)   # all slices of a step
        roof["kernel_ms_per_launch"] = kernel_ms
        res = {
            "metric": "QPS @ recall@10>=0.999, 10Mx768 L2",
            "value": qps, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "recall_at_10": recall,
            "config": {"workload": "%dM x %d L2 flat/ANN search, k=%d, batch=%d per step, %s rows per GPU, %d GPU(s), "
                                   "rows_total=%d%s" % ((n * world) // 1_000_000, d, k, b, n, world, n * world,
                                                        "" if world == 1 else (" (corpus hash-sharded over the GPUs, batch = %d x %d: per-GPU work fixed)" % (world, args.batch)
                                                                               if args.scale == "queries" else " (rows per GPU fixed, same batch: capacity scaling, see work_rate)")),
                       "mode": args.mode, "engine": args.engine, "parallelism": "row-hash-shard x%d + RCCL all-gather top-k" % world},
            "roofline": roof,
            "stats": {"dist_evals_per_query": st["dist_evals"] / float(b), "rerank_rows_per_query": st["rerank_rows"] / float(b),
                      "overflow_queries": st["overflow_queries"]},
            "work_rate": {"value": qps * n * world, "unit": "query*rows/s"},
        }
        if args.cpu_seconds > 0 and world == 1:
            try:
                res["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
            except Exception as e:  # the baseline is a report, never a dependency of the product path
                res["cpu_baseline"] = {"value": None, "unit": "queries/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
