#!/usr/bin/env python
"""p50 of a single query under a compiled filter program (id % 3 = 1) at 1M x 768 on the one-pass form:  python filter_program_p50.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vectordb_amd as amd  # noqa: E402

n, d = 1_000_000, 768
g = torch.Generator(device="cuda").manual_seed(42)
X = torch.rand((n, d), generator=g, device="cuda")
Q = torch.rand((64, d), generator=g, device="cuda")
ix = amd.GpuIndex(d, 0).use_torch_stream()
ix.attach_rows(X)
attr = torch.arange(n, dtype=torch.int32, device="cuda").view(torch.uint8).reshape(n, 4)
o = (torch.empty((1, 10), dtype=torch.int64, device="cuda"), torch.empty((1, 10), device="cuda"), torch.empty((1,), dtype=torch.int32, device="cuda"))
for what, prog in (("no filter", None), ("id % 3 = 1", [("i32", 0), ("const", 3), ("%",), ("const", 1), ("=",)])):
    ix.set_filter_program(prog, attr if prog else None, stride=4)
    lat, one = [], 0
    for i in range(60):
        t0 = time.perf_counter()
        ix.search(Q[i % 64:i % 64 + 1], 10, out=o, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t0)
        one += ix.stats()["one_pass"]
    print("%-12s p50 %.3f ms (one-pass calls %d of 60; the last 40 timed)" % (what, 1e3 * float(np.median(lat[20:])), one), flush=True)
