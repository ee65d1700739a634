#!/usr/bin/env python
"""rocprofv3 driver for the traversal kernel: corpus + graph (loaded from a file written by bench_graph.py/bench.py),
ONE fp32 stream scan of 4 queries (a launch whose HBM bytes are known exactly, rows*dim*4, with the same 16 B/lane
row-load pattern as the traversal: the calibration point for FETCH_SIZE), then traversal launches.
    python scripts/prof_graph.py GRAPH.bin [rows] [dim] [batch] [iters]       -> one JSON line per launch"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("EPS_TUNING_FROM_ENV", "1")   # (scripts steer the library's engine switches through the environment: vectordb_amd/_lib.py)
import vectordb_amd as amd  # noqa: E402

path = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
d = int(sys.argv[3]) if len(sys.argv) > 3 else 768
b = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 2
g = torch.Generator(device="cuda").manual_seed(42)
X = torch.empty((n, d), device="cuda")
for s in range(0, n, 1 << 19):   # same slabs / seed as bench.py and bench_graph.py: the graph file matches these rows
    e = min(n, s + (1 << 19))
    X[s:e] = torch.rand((e - s, d), generator=g, device="cuda")
Q = torch.rand((b, d), generator=torch.Generator(device="cuda").manual_seed(43), device="cuda")
ix = amd.GpuIndex(d, 0).use_torch_stream()
ix.attach_rows(X)
ix.load_graph(path)
n_, e_, nav = ix.graph_info()
out = (torch.empty((b, 10), dtype=torch.int64, device="cuda"), torch.empty((b, 10), device="cuda"), torch.empty((b,), dtype=torch.int32, device="cuda"))
ix.search(Q[:4], 10, out=tuple(o[:4] for o in out), mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
torch.cuda.synchronize()
st = ix.stats()
print(json.dumps({"launch": "flat_scan_kernel 4 queries (calibration)", "bytes_exact": n * d * 4, "kernel_ms": st["main_kernel_ms"]}))
for T in (4, 1):
    for it in range(iters):
        ix.search(Q, 10, out=out, mode=amd.MODE_GRAPH, intra_threads=T, master_queue=500, local_queue=500)
        torch.cuda.synchronize()
        st = ix.stats()
        alg = amd.traversal_gather_bytes(st, d, e_ / float(n_), 500 * b)
        print(json.dumps({"launch": "traverse2_kernel T=%d L=500 batch=%d #%d" % (T, b, it), "dist_evals": st["dist_evals"], "expansions": st["expansions"],
                          "algorithmic_bytes": alg, "kernel_ms": st["main_kernel_ms"], "algorithmic_GBps": alg / (st["main_kernel_ms"] * 1e-3) / 1e9}))
