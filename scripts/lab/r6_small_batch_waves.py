"""r6: wavefronts per query for a handful of queries (the library takes 16 up to 256 queries): p50 per call by table shape and EPS_TRV_WAVES; T = 4 / 1, L = 500."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
os.environ["EPS_TUNING_FROM_ENV"] = "1"
import vectordb_amd as amd
for n, d in ((100_000, 128), (1_000_000, 768)):
    g = torch.Generator(device="cuda").manual_seed(42)
    X = torch.rand((n, d), generator=g, device="cuda")
    Q = torch.rand((256, d), generator=g, device="cuda")
    ix = amd.GpuIndex(d, 0).use_torch_stream(); ix.attach_rows(X); ix.build(); ix.synchronize()
    for T in (4, 1):
        for nq in (1, 16):
            o = (torch.empty((nq, 10), dtype=torch.int64, device="cuda"), torch.empty((nq, 10), device="cuda"), torch.empty((nq,), dtype=torch.int32, device="cuda"))
            line = "%8d x %4d  T %d  %2d queries per call:" % (n, d, T, nq)
            for w in ("4", "8", "16"):
                os.environ["EPS_TRV_WAVES"] = w
                kw = dict(mode=amd.MODE_GRAPH, intra_threads=T, master_queue=500, local_queue=500)
                for i in range(3):
                    ix.search(Q[i:i + nq], 10, out=o, **kw)
                torch.cuda.synchronize()
                lat = []
                for i in range(40):
                    t0 = time.perf_counter(); ix.search(Q[i:i + nq], 10, out=o, **kw); torch.cuda.synchronize(); lat.append(time.perf_counter() - t0)
                line += "  %2s wavefronts p50 %.3f ms" % (w, 1e3 * float(np.median(lat)))
            del os.environ["EPS_TRV_WAVES"]
            print(line, flush=True)
    ix.close(); del X
