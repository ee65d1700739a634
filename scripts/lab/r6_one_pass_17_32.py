"""r6: 17..32 queries per call on the one-pass form (two 16-query column blocks, stream8m_kernel<., 2>) against the staged chain: p50 per call, issue -> sync,
1M x 768 uniform rows (BASELINE configs[1]'s table) and the embedding-like table; answers against the stream engine."""
import os
import sys
import time

import numpy as np
import torch

os.environ["EPS_TUNING_FROM_ENV"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vectordb_amd as amd  # noqa: E402

n, d, k = 1_000_000, 768, 10
dev = torch.device("cuda", 0)
for kind in ("uniform", "embedding"):
    g = torch.Generator(device=dev).manual_seed(42)
    if kind == "uniform":
        X = torch.rand((n, d), generator=g, device=dev)
        Q = torch.rand((64, d), generator=g, device=dev)
        metric = "EUCLIDEAN"
    else:
        scale = torch.ones((d,), device=dev)
        scale[:8] = 4.0
        X = torch.randn((n, d), generator=g, device=dev) * scale
        Q = torch.randn((64, d), generator=g, device=dev) * scale
        amd.normalize_rows(X, only_if_nonzero=True, device=0, stream=torch.cuda.current_stream().cuda_stream)
        amd.normalize_rows(Q, only_if_nonzero=False, device=0, stream=torch.cuda.current_stream().cuda_stream)
        metric = "COSINE"
    torch.cuda.synchronize()
    ix = amd.GpuIndex(d, metric, device=0).use_torch_stream()
    ix.attach_rows(X)
    for nq in (8, 16, 17, 24, 32):
        o = (torch.empty((nq, k), dtype=torch.int64, device=dev), torch.empty((nq, k), device=dev), torch.empty((nq,), dtype=torch.int32, device=dev))
        r = (torch.empty((nq, k), dtype=torch.int64, device=dev), torch.empty((nq, k), device=dev), torch.empty((nq,), dtype=torch.int32, device=dev))
        ix.search(Q[:nq], k, out=r, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
        torch.cuda.synchronize()
        line = "%s %2d queries per call:" % (kind, nq)
        for name, maxq in (("one-pass", "32"), ("staged chain", "16" if nq > 16 else "4")):
            os.environ["EPS_S8_MAX_Q"] = maxq
            lat, one = [], 0
            for i in range(60):
                t0 = time.perf_counter()
                ix.search(Q[i % 8:i % 8 + nq], k, out=o, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
                torch.cuda.synchronize()
                lat.append(time.perf_counter() - t0)
                one += ix.stats()["one_pass"]
            ix.search(Q[:nq], k, out=o, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
            torch.cuda.synchronize()
            same = bool(torch.equal(o[0], r[0]) and torch.equal(o[1], r[1]))
            line += "  %s p50 %.3f ms (one-pass calls %d of 60, == scan %s)" % (name, 1e3 * float(np.median(lat[10:])), one, same)
        del os.environ["EPS_S8_MAX_Q"]
        print(line, flush=True)
    ix.close()
    del X
