"""Stream vs MFMA flat engine at small / medium batch sizes (where should FLAT_AUTO switch?).
    python scripts/bench_midbatch.py [rows] [dim]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("EPS_TUNING_FROM_ENV", "1")   # (scripts steer the library's engine switches through the environment: vectordb_amd/_lib.py)
import vectordb_amd as amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
g = torch.Generator(device="cuda").manual_seed(42)
X = torch.empty((n, d), device="cuda")
for s in range(0, n, 1 << 20):
    e = min(n, s + (1 << 20))
    X[s:e] = torch.rand((e - s, d), generator=g, device="cuda")
ix = amd.GpuIndex(d, 0).use_torch_stream()
ix.attach_rows(X)
for nq in (1, 2, 4, 8, 16, 32, 64, 128):
    Q = torch.rand((nq, d), generator=g, device="cuda")
    out = (torch.empty((nq, 10), dtype=torch.int64, device="cuda"), torch.empty((nq, 10), device="cuda"), torch.empty((nq,), dtype=torch.int32, device="cuda"))
    res = {}
    for name, eng in (("stream", amd.FLAT_STREAM), ("mfma", amd.FLAT_MFMA)):
        for _ in range(2):
            ix.search(Q, 10, out=out, mode=amd.MODE_FLAT, flat_engine=eng)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            ix.search(Q, 10, out=out, mode=amd.MODE_FLAT, flat_engine=eng)
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / reps * 1e3
    print("n=%d d=%d nq=%4d  stream %8.3f ms   mfma %8.3f ms" % (n, d, nq, res["stream"], res["mfma"]), flush=True)
