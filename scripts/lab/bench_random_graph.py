#!/usr/bin/env python
"""Gather-bandwidth proxy for the traversal kernel at full table size WITHOUT the minutes-long graph build: a random regular
graph (every node points at `deg` uniformly random rows) gives the kernel the same access pattern as the NSG does on uniform
data - random 4*d-byte row gathers over the whole table - so kernel variants (EPS_TRV_WAVES, EPS_TRV_PER_CU, batch) can be
compared in seconds.  Results are not recall figures.
    python scripts/lab/bench_random_graph.py [rows] [dim] [deg] [batches e.g. 1024,2048]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("EPS_TUNING_FROM_ENV", "1")   # (scripts steer the library's engine switches through the environment: vectordb_amd/_lib.py)
import vectordb_amd as amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
deg = int(sys.argv[3]) if len(sys.argv) > 3 else 48
batches = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "1024").split(",")]
g = torch.Generator(device="cuda").manual_seed(42)
X = torch.empty((n, d), device="cuda")
for s in range(0, n, 1 << 19):
    e = min(n, s + (1 << 19))
    X[s:e] = torch.rand((e - s, d), generator=g, device="cuda")
t0 = time.time()
nbr = torch.randint(0, n, (n, deg), generator=torch.Generator(device="cuda").manual_seed(7), device="cuda", dtype=torch.int64).cpu().numpy().reshape(-1)
off = np.arange(0, (n + 1) * deg, deg, dtype=np.int64)
ix = amd.GpuIndex(d, 0).use_torch_stream()
ix.attach_rows(X)
ix.set_graph(off, nbr, 0)
print("graph ready in %.1f s" % (time.time() - t0), file=sys.stderr)
for b in batches:
    Q = torch.rand((b, d), generator=torch.Generator(device="cuda").manual_seed(43), device="cuda")
    out = (torch.empty((b, 10), dtype=torch.int64, device="cuda"), torch.empty((b, 10), device="cuda"), torch.empty((b,), dtype=torch.int32, device="cuda"))
    for T in (1, 4):
        for waves in (os.environ.get("WAVES", "4,8").split(",")):
            if waves == "auto":
                os.environ.pop("EPS_TRV_WAVES", None)
            else:
                os.environ["EPS_TRV_WAVES"] = waves
            LQ = int(os.environ.get("LQ", "500"))
            kw = dict(mode=amd.MODE_GRAPH, intra_threads=T, master_queue=LQ, local_queue=LQ)
            ix.search(Q, 10, out=out, **kw)
            ms = []
            for _ in range(3):
                ix.search(Q, 10, out=out, **kw)
                torch.cuda.synchronize()
                ms.append(ix.stats()["main_kernel_ms"])
            st = ix.stats()
            alg = st["dist_evals"] * (4.0 * d + 4) + st["expansions"] * (8 + 4.0 * deg)
            km = float(np.median(ms))
            print(json.dumps({"rows": n, "batch": b, "T": T, "waves_per_query": waves, "per_cu": os.environ.get("EPS_TRV_PER_CU", "auto"),
                              "kernel_ms": km, "evals_per_query": st["dist_evals"] / b, "GBps": alg / (km * 1e-3) / 1e9, "frac_of_8TBps": alg / (km * 1e-3) / 8e12}), flush=True)
