"""Throughput of the exact flat scan (FLAT_AUTO) against the batch size at the headline table (JSON lines):
    python scripts/bench_batch_sweep.py [rows=10000000] [dim=768]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("EPS_TUNING_FROM_ENV", "1")   # (scripts steer the library's engine switches through the environment: vectordb_amd/_lib.py)
import vectordb_amd as amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
g = torch.Generator(device="cuda").manual_seed(42)
X = torch.empty((n, d), device="cuda")
for s in range(0, n, 1 << 20):
    e = min(n, s + (1 << 20))
    X[s:e] = torch.rand((e - s, d), generator=g, device="cuda")
ix = amd.GpuIndex(d, 0).use_torch_stream()
ix.attach_rows(X)
# (FLAT_AUTO builds the 8-bit mirror only when a table keeps getting calls - 17 small ones, or the first large one; build it up front so that no timed call pays for it)
_q = torch.rand((64, d), generator=g, device="cuda")
ix.search(_q, 10, out=(torch.empty((64, 10), dtype=torch.int64, device="cuda"), torch.empty((64, 10), device="cuda"), torch.empty((64,), dtype=torch.int32, device="cuda")),
          mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
torch.cuda.synchronize()
for nq in (1, 4, 8, 16, 24, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192):
    Q = torch.rand((nq, d), generator=g, device="cuda")
    out = (torch.empty((nq, 10), dtype=torch.int64, device="cuda"), torch.empty((nq, 10), device="cuda"), torch.empty((nq,), dtype=torch.int32, device="cuda"))
    for _ in range(2):
        ix.search(Q, 10, out=out, mode=amd.MODE_FLAT)
    torch.cuda.synchronize()
    reps = 8 if nq <= 2048 else 3
    t0 = time.perf_counter()
    for _ in range(reps):
        ix.search(Q, 10, out=out, mode=amd.MODE_FLAT)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    st = ix.stats()
    print(json.dumps({"config": "%d x %d L2 exact flat scan (FLAT_AUTO), k=10" % (n, d), "batch": nq, "ms_per_call": ms, "qps": nq / ms * 1e3,
                      "engine": "mfma filter + re-rank" if st["rerank_rows"] > 0 else "fp32 stream scan", "rerank_rows_per_query": st["rerank_rows"] / float(nq), "one_pass": int(st.get("one_pass", 0))}), flush=True)
