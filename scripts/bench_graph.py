#!/usr/bin/env python
"""Graph-traversal measurements: device build, then a sweep of SearchQueueSize / IntraQueryThreads with
recall@10 against the exact scan, queries/s, distance evaluations and the algorithmic gather rate
(vectordb_amd.traversal_gather_bytes) over the kernel time.

    python scripts/bench_graph.py --rows 1000000 --dim 768 --data uniform|clustered [--batch 1024] [--L 500,2000]
                                  [--T 1,4] [--save-graph PATH | --load-graph PATH]        -> JSON lines

data = uniform: i.i.d. U[0,1) (the BASELINE recipe, adversarial for any graph index);
data = clustered: the SURVEY 8d secondary set, 1000 Gaussian clusters (centres U[0,1)^d, sigma = 0.1), queries from
the same mixture."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("EPS_TUNING_FROM_ENV", "1")   # (scripts steer the library's engine switches through the environment: vectordb_amd/_lib.py)
import vectordb_amd as amd  # noqa: E402


def gen(n, d, seed, kind, centres=None):
    g = torch.Generator(device="cuda").manual_seed(seed)
    X = torch.empty((n, d), device="cuda")
    step = 1 << 19
    for s in range(0, n, step):
        e = min(n, s + step)
        if kind == "uniform":
            X[s:e] = torch.rand((e - s, d), generator=g, device="cuda")
        elif kind == "manifold":   # centres = (A [16, d], noise sigma): intrinsic dimension 16 embedded in d
            z = torch.rand((e - s, centres.shape[0]), generator=g, device="cuda")
            X[s:e] = z @ centres + 0.01 * torch.randn((e - s, d), generator=g, device="cuda")
        else:
            a = torch.randint(0, centres.shape[0], (e - s,), generator=g, device="cuda")
            X[s:e] = centres[a] + 0.1 * torch.randn((e - s, d), generator=g, device="cuda")
    return X


def outs(nq, k):
    return (torch.empty((nq, k), dtype=torch.int64, device="cuda"), torch.empty((nq, k), dtype=torch.float32, device="cuda"),
            torch.empty((nq,), dtype=torch.int32, device="cuda"))


def recall(a, b):
    return float(np.mean([len(set(x) & set(y)) / float(len(y)) for x, y in zip(a, b)]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--data", default="uniform", choices=["uniform", "clustered", "manifold"],
                    help="manifold: a 16-dimensional uniform latent embedded linearly in --dim dimensions + 1 %% noise (low intrinsic dimension, "
                         "as learned embeddings have; not part of the BASELINE recipe)")
    ap.add_argument("--L", default="500,2000")
    ap.add_argument("--T", default="1,4")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--save-graph", default=None)
    ap.add_argument("--load-graph", default=None)
    ap.add_argument("--save-data", default=None, help="directory: rows.f32 / queries.f32 / gt.i64 for the CPU legs")
    args = ap.parse_args()
    n, d, b, k = args.rows, args.dim, args.batch, args.k
    centres = None
    if args.data == "clustered":
        centres = torch.rand((1000, d), generator=torch.Generator(device="cuda").manual_seed(41), device="cuda")
    if args.data == "manifold":
        centres = 0.25 * torch.randn((16, d), generator=torch.Generator(device="cuda").manual_seed(41), device="cuda")
    X = gen(n, d, 42, args.data, centres)
    Q = gen(b, d, 43, args.data, centres)
    ix = amd.GpuIndex(d, 0).use_torch_stream()
    ix.attach_rows(X)
    gt = outs(b, k)
    ix.search(Q, k, out=gt, mode=amd.MODE_FLAT)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ix.search(Q, k, out=gt, mode=amd.MODE_FLAT)
    torch.cuda.synchronize()
    flat_s = time.perf_counter() - t0
    gti = gt[0].cpu().numpy()
    print(json.dumps({"config": "%s %d x %d, exact flat scan batch=%d" % (args.data, n, d, b), "qps": b / flat_s, "recall_at_10": 1.0}), flush=True)
    t0 = time.perf_counter()
    if args.load_graph:
        ix.load_graph(args.load_graph)
    else:
        ix.build()
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    n_, e_, nav = ix.graph_info()
    print(json.dumps({"config": "%s %d x %d graph %s" % (args.data, n, d, "load" if args.load_graph else "build (kNN K=100 on MFMA + NSG)"),
                      "seconds": build_s, "avg_degree": e_ / n_, "nav": nav}), flush=True)
    if args.save_graph:
        ix.save_graph(args.save_graph)
    if args.save_data:
        os.makedirs(args.save_data, exist_ok=True)
        X.cpu().numpy().tofile(os.path.join(args.save_data, "rows.f32"))
        Q.cpu().numpy().tofile(os.path.join(args.save_data, "queries.f32"))
        gti.tofile(os.path.join(args.save_data, "gt.i64"))
    deg = e_ / float(n_)
    ob = outs(b, k)
    for T in [int(x) for x in args.T.split(",")]:
        for L in [int(x) for x in args.L.split(",")]:
            kw = dict(mode=amd.MODE_GRAPH, intra_threads=T, master_queue=L, local_queue=L)
            ix.search(Q, k, out=ob, **kw)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            kms = []
            for _ in range(args.reps):
                ix.search(Q, k, out=ob, **kw)
                kms.append(ix.stats()["main_kernel_ms"])
            torch.cuda.synchronize()
            el = (time.perf_counter() - t1) / args.reps
            st = ix.stats()
            byt = amd.traversal_gather_bytes(st, d, deg, min(L, n) * b)
            km = float(np.median(kms))
            print(json.dumps({"config": "%s %d x %d graph traversal batch=%d T=%d L=%d" % (args.data, n, d, b, T, L),
                              "qps": b / el, "recall_at_10": recall(ob[0].cpu().numpy(), gti), "evals_per_query": st["dist_evals"] / float(b),
                              "expansions_per_query": st["expansions"] / float(b), "fp32_rows_per_query": st["rerank_rows"] / float(b), "kernel_ms": km,
                              "achieved_GBps": byt / (km * 1e-3) / 1e9, "frac_of_8TBps": byt / (km * 1e-3) / 8e12,
                              "vs_flat_qps": (b / el) / (b / flat_s)}), flush=True)


if __name__ == "__main__":
    main()
