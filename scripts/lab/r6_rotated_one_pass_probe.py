#!/usr/bin/env python
"""r6: does the one-pass search serve an embedding-like table in the rotated frame?  [rows] x 768 unit-norm rows with 8 dominant columns, COSINE; 1 / 3 / 8 / 16
queries per call; prints what the call took (one pass or the staged chain), whether per-row margins were folded, and the p50 over 200 calls."""
import os
import sys
import time

import numpy as np
import torch

os.environ.setdefault("EPS_TUNING_FROM_ENV", "1")
os.environ.setdefault("EPS_DEBUG_ONE_PASS_OVERFLOW", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vectordb_amd as amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d, k = 768, 10
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(77)
scale = torch.ones((d,), device=dev)
scale[:8] = 4.0
X = torch.empty((n, d), device=dev)
for s in range(0, n, 1 << 19):
    e = min(n, s + (1 << 19))
    X[s:e] = torch.randn((e - s, d), generator=g, device=dev) * scale
amd.normalize_rows(X, only_if_nonzero=True, device=0, stream=torch.cuda.current_stream().cuda_stream)
Q = torch.randn((64, d), generator=torch.Generator(device=dev).manual_seed(78), device=dev) * scale
amd.normalize_rows(Q, only_if_nonzero=False, device=0, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
for rot in (None, "0"):
    if rot is None:
        os.environ.pop("EPS_MIRROR_ROTATE", None)
    else:
        os.environ["EPS_MIRROR_ROTATE"] = rot
    ix = amd.GpuIndex(d, "COSINE", device=0).use_torch_stream()
    ix.attach_rows(X)
    for nq in (1, 3, 8, 16):
        o = (torch.empty((nq, k), dtype=torch.int64, device=dev), torch.empty((nq, k), device=dev), torch.empty((nq,), dtype=torch.int32, device=dev))
        ts = []
        for rep in range(200):
            q = Q[(rep * nq) % 48:(rep * nq) % 48 + nq]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ix.search(q, k, out=o, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            if rep == 0:
                st = ix.stats()
        st_last = ix.stats()
        o2 = (torch.empty((nq, k), dtype=torch.int64, device=dev), torch.empty((nq, k), device=dev), torch.empty((nq,), dtype=torch.int32, device=dev))
        ix.search(q, k, out=o2, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
        torch.cuda.synchronize()
        ok = bool((o2[0] == o[0]).all()) and bool((o2[1] == o[1]).all())
        print("frame", "auto" if rot is None else "identity", "queries", nq, "first call: one_pass", st["one_pass"], "rotated", st["i8_rotated"], "folded", st["i8_folded"], "bits", st["main_kernel_bits"],
              "rerank rows/query %.0f" % (st["rerank_rows"] / nq), "| last call one_pass", st_last["one_pass"], "p50 %.3f ms" % (1e3 * float(np.median(ts))), "== scan", ok, flush=True)
    ix.close()
