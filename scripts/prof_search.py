#!/usr/bin/env python
"""Small driver for rocprofv3 runs: builds the bench corpus and runs a few searches, nothing else on the GPU.
    python scripts/prof_search.py [rows] [dim] [batch] [iters] [engine: auto|stream|mfma]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("EPS_TUNING_FROM_ENV", "1")   # (scripts steer the library's engine switches through the environment: vectordb_amd/_lib.py)
import vectordb_amd as amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
b = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 2
eng = {"auto": amd.FLAT_AUTO, "stream": amd.FLAT_STREAM, "mfma": amd.FLAT_MFMA}[sys.argv[5] if len(sys.argv) > 5 else "auto"]
g = torch.Generator(device="cuda").manual_seed(42)
X = torch.empty((n, d), device="cuda")
for s in range(0, n, 1 << 20):
    e = min(n, s + (1 << 20))
    X[s:e] = torch.rand((e - s, d), generator=g, device="cuda")
Q = torch.rand((b, d), generator=g, device="cuda")
ix = amd.GpuIndex(d, 0).use_torch_stream()
ix.attach_rows(X)
out = (torch.empty((b, 10), dtype=torch.int64, device="cuda"), torch.empty((b, 10), device="cuda"), torch.empty((b,), dtype=torch.int32, device="cuda"))
for _ in range(iters):
    ix.search(Q, 10, out=out, mode=amd.MODE_FLAT, flat_engine=eng)
torch.cuda.synchronize()
print(ix.stats())
