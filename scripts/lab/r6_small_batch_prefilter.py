"""r6: does the traversal's 8-bit prefilter pay for a HANDFUL of queries?  It saves bytes, and a single query is latency-bound: the prefilter is one more dependent
round trip per step (mirror row, then the survivor's fp32 row).  p50 per call by table shape, queries per call and EPS_TRV_PREFILTER; T = 4, L = 500."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.getcwd())
os.environ["EPS_TUNING_FROM_ENV"] = "1"
import vectordb_amd as amd
for n, d in ((100_000, 128), (1_000_000, 128), (1_000_000, 384), (1_000_000, 768)):
    g = torch.Generator(device="cuda").manual_seed(42)
    X = torch.rand((n, d), generator=g, device="cuda")
    Q = torch.rand((256, d), generator=g, device="cuda")
    ix = amd.GpuIndex(d, 0).use_torch_stream(); ix.attach_rows(X); ix.build(); ix.synchronize()
    for nq in (1, 4, 16, 64):
        o = (torch.empty((nq, 10), dtype=torch.int64, device="cuda"), torch.empty((nq, 10), device="cuda"), torch.empty((nq,), dtype=torch.int32, device="cuda"))
        line = "%8d x %4d  %3d queries per call:" % (n, d, nq)
        for pf in ("1", "0"):
            os.environ["EPS_TRV_PREFILTER"] = pf
            kw = dict(mode=amd.MODE_GRAPH, intra_threads=4, master_queue=500, local_queue=500)
            for i in range(3):
                ix.search(Q[i:i + nq], 10, out=o, **kw)
            torch.cuda.synchronize()
            lat = []
            for i in range(60):
                t0 = time.perf_counter(); ix.search(Q[i:i + nq], 10, out=o, **kw); torch.cuda.synchronize(); lat.append(time.perf_counter() - t0)
            line += "  prefilter %s p50 %.3f ms" % (pf, 1e3 * float(np.median(lat)))
        del os.environ["EPS_TRV_PREFILTER"]
        print(line, flush=True)
    ix.close(); del X
