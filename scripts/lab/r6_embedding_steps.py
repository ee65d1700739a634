#!/usr/bin/env python
"""r6: N back-to-back steps of the exact scan on the embedding-like table (10M x 768 COSINE, unit-norm Gaussian rows, 8 dominant columns: bench.py's
`embedding_like` leg) - for rocprofv3 --kernel-trace --stats (where the step's time goes) and for the sustained rate."""
import os
import sys
import time

import torch

os.environ.setdefault("EPS_TUNING_FROM_ENV", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vectordb_amd as amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
d, b, k = 768, 1024, 10
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(77)
scale = torch.ones((d,), device=dev)
scale[:8] = 4.0
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
for _ in range(3):
    ix.search(Q, k, out=o, mode=amd.MODE_FLAT)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    ix.search(Q, k, out=o, mode=amd.MODE_FLAT)
torch.cuda.synchronize()
sec = (time.perf_counter() - t0) / steps
st = ix.stats()
print("embedding-like %d x %d: %.3f ms/step = %.1f k q/s, bits %d rotated %d rerank rows/query %.0f overflow %d main kernel ms %s" % (
    n, d, 1e3 * sec, b / sec / 1e3, st["main_kernel_bits"], st.get("i8_rotated", -1), st["rerank_rows"] / b, st["overflow_queries"], ix.kernel_times(4)), flush=True)
if os.environ.get("EPS_DEBUG_ONE"):
    os.environ["EPS_DEBUG"] = "1"
    ix.search(Q, k, out=o, mode=amd.MODE_FLAT)
    torch.cuda.synchronize()
