#!/usr/bin/env python
"""p50 latency of the 8-bit filter path by batch size under the current EPS_MFMA_STAGES:  python stages_by_batch.py rows dim"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("EPS_TUNING_FROM_ENV", "1")   # (scripts steer the library's engine switches through the environment: vectordb_amd/_lib.py)
import vectordb_amd as amd  # noqa: E402

n, d = int(sys.argv[1]), int(sys.argv[2])
g = torch.Generator(device="cuda").manual_seed(42)
X = torch.empty((n, d), device="cuda")
for s in range(0, n, 1 << 19):
    e = min(n, s + (1 << 19))
    X[s:e] = torch.rand((e - s, d), generator=g, device="cuda")
Q = torch.rand((4096, d), generator=g, device="cuda")
ix = amd.GpuIndex(d, 0).use_torch_stream()
ix.attach_rows(X)
out = []
for nq in (1, 2, 4, 5, 8, 16, 32, 64, 128, 256):
    o = (torch.empty((nq, 10), dtype=torch.int64, device="cuda"), torch.empty((nq, 10), device="cuda"), torch.empty((nq,), dtype=torch.int32, device="cuda"))
    ix.search(Q[:nq], 10, out=o, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
    torch.cuda.synchronize()
    lat = []
    for i in range(24):
        q = Q[(i * nq) % 2048:(i * nq) % 2048 + nq]
        t0 = time.perf_counter()
        ix.search(q, 10, out=o, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t0)
    out.append("%d:%.3f" % (nq, 1e3 * float(np.median(lat))))
print("stages", os.environ.get("EPS_MFMA_STAGES", "default"), "rows", n, " ".join(out))
