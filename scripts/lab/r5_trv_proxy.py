#!/usr/bin/env python
"""Traversal kernel variants at full table size without the graph build (r5): 10M x 768 uniform rows + a random regular graph (the same
access pattern as the NSG on uniform data: random row gathers over the whole table).  Kernel time per (T, L), the evaluation counts
(they must agree between variants: the walk does not depend on the variant) and a checksum of the returned ids.
    python scripts/lab/r5_trv_proxy.py [rows] [dim] [deg] [T:L,T:L,...]"""
import json
import os
import sys
import time
import zlib

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("EPS_TUNING_FROM_ENV", "1")   # (scripts steer the library's engine switches through the environment: vectordb_amd/_lib.py)
import vectordb_amd as amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
deg = int(sys.argv[3]) if len(sys.argv) > 3 else 48
cases = [tuple(int(x) for x in c.split(":")) for c in (sys.argv[4] if len(sys.argv) > 4 else "4:500,1:500,1:100,4:100").split(",")]
b = int(os.environ.get("BATCH", "1024"))
g = torch.Generator(device="cuda").manual_seed(42)
X = torch.empty((n, d), device="cuda")
for s in range(0, n, 1 << 19):
    e = min(n, s + (1 << 19))
    X[s:e] = torch.rand((e - s, d), generator=g, device="cuda")
gf = "/tmp/r5_proxy_graph_%d_%d.npy" % (n, deg)
if os.path.exists(gf):
    nbr = np.load(gf)
else:
    nbr = torch.randint(0, n, (n, deg), generator=torch.Generator(device="cuda").manual_seed(7), device="cuda", dtype=torch.int64).cpu().numpy().reshape(-1)
    np.save(gf, nbr)
off = np.arange(0, (n + 1) * deg, deg, dtype=np.int64)
ix = amd.GpuIndex(d, 0).use_torch_stream()
ix.attach_rows(X)
ix.set_graph(off, nbr, 0)
Q = torch.rand((b, d), generator=torch.Generator(device="cuda").manual_seed(43), device="cuda")
out = (torch.empty((b, 10), dtype=torch.int64, device="cuda"), torch.empty((b, 10), device="cuda"), torch.empty((b,), dtype=torch.int32, device="cuda"))
for T, L in cases:
    kw = dict(mode=amd.MODE_GRAPH, intra_threads=T, master_queue=L, local_queue=L)
    ix.search(Q, 10, out=out, **kw)
    ms = []
    for _ in range(int(os.environ.get("REPS", "5"))):
        ix.search(Q, 10, out=out, **kw)
        torch.cuda.synchronize()
        ms.append(ix.stats()["main_kernel_ms"])
    st = ix.stats()
    km = float(np.median(ms))
    alg = amd.traversal_gather_bytes(st, d, deg, seed_evals=L * b)
    print(json.dumps({"variant": os.environ.get("VARIANT", "?"), "T": T, "L": L, "batch": b, "kernel_ms": round(km, 4), "min_ms": round(min(ms), 4),
                      "evals_per_query": st["dist_evals"] / b, "fp32_rows_per_query": st["rerank_rows"] / b, "expansions_per_query": st["expansions"] / b,
                      "ids_crc": zlib.crc32(out[0].cpu().numpy().tobytes()), "qps": round(b / (km * 1e-3)), "frac_of_8TBps": round(alg / (km * 1e-3) / 8e12, 4)}), flush=True)
