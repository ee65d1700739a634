#!/usr/bin/env python
"""Why does the embedding-like set (unit-norm Gaussian rows, 8 dominant dimensions) leave the 8-bit first pass?  The flat engine's choices with EPS_DEBUG
on, at [rows] x 768 COSINE, batch 1024: forced 8-bit, forced fp16, auto."""
import os
import sys
import time

import numpy as np
import torch

os.environ.setdefault("EPS_TUNING_FROM_ENV", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vectordb_amd as amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
d, b, k = 768, 1024, 10
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(77)
scale = torch.ones((d,), device=dev)
scale[:8] = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
X = torch.empty((n, d), device=dev)
for s in range(0, n, 1 << 19):
    e = min(n, s + (1 << 19))
    X[s:e] = torch.randn((e - s, d), generator=g, device=dev) * scale
amd.normalize_rows(X, only_if_nonzero=True, device=0, stream=torch.cuda.current_stream().cuda_stream)
Q = torch.randn((b, d), generator=torch.Generator(device=dev).manual_seed(78), device=dev) * scale
amd.normalize_rows(Q, only_if_nonzero=False, device=0, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
ix = amd.GpuIndex(d, "COSINE", device=0).use_torch_stream()
ix.attach_rows(X)
o = (torch.empty((b, k), dtype=torch.int64, device=dev), torch.empty((b, k), device=dev), torch.empty((b,), dtype=torch.int32, device=dev))
for eng, name in ((amd.FLAT_MFMA_I8, "int8"), (amd.FLAT_MFMA, "fp16"), (amd.FLAT_AUTO, "auto")):
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ix.search(Q, k, out=o, mode=amd.MODE_FLAT, flat_engine=eng)
        torch.cuda.synchronize()
        st = ix.stats()
        print(name, rep, "ms %.3f" % (1e3 * (time.perf_counter() - t0)), "bits", st["main_kernel_bits"], "rerank rows/query %.0f" % (st["rerank_rows"] / b), "overflow", st["overflow_queries"], flush=True)
