"""Timeline of ONE single-query exact search (8-bit filter) on rows x dim: python scripts/prof_single_query.py [rows] [dim] under rocprofv3 --kernel-trace"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("EPS_TUNING_FROM_ENV", "1")   # (scripts steer the library's engine switches through the environment: vectordb_amd/_lib.py)
import vectordb_amd as amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
g = torch.Generator(device="cuda").manual_seed(42)
X = torch.rand((n, d), generator=g, device="cuda")
Q = torch.rand((64, d), generator=g, device="cuda")
ix = amd.GpuIndex(d, 0).use_torch_stream()
ix.attach_rows(X)
o = (torch.empty((1, 10), dtype=torch.int64, device="cuda"), torch.empty((1, 10), device="cuda"), torch.empty((1,), dtype=torch.int32, device="cuda"))
o64 = (torch.empty((64, 10), dtype=torch.int64, device="cuda"), torch.empty((64, 10), device="cuda"), torch.empty((64,), dtype=torch.int32, device="cuda"))
ix.search(Q, 10, out=o64, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
torch.cuda.synchronize()
lat = []
for i in range(40):
    t0 = time.perf_counter()
    ix.search(Q[i:i + 1], 10, out=o, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
    torch.cuda.synchronize()
    lat.append(time.perf_counter() - t0)
print("p50 ms", 1e3 * float(np.median(lat)))
