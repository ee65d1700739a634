"""The one-pass search at 10M x 768: three single-vector calls against the same rows of a 64-query batch (bit for bit) + their p50."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("EPS_TUNING_FROM_ENV", "1")   # (scripts steer the library's engine switches through the environment: vectordb_amd/_lib.py)
import vectordb_amd as amd  # noqa: E402

n, d, k = 10_000_000, 768, 10
g = torch.Generator(device="cuda").manual_seed(42)
X = torch.empty((n, d), device="cuda")
for s in range(0, n, 1 << 19):
    e = min(n, s + (1 << 19))
    X[s:e] = torch.rand((e - s, d), generator=g, device="cuda")
Q = torch.rand((64, d), generator=g, device="cuda")
ix = amd.GpuIndex(d, 0).use_torch_stream()
ix.attach_rows(X)


def outs(b):
    return (torch.empty((b, k), dtype=torch.int64, device="cuda"), torch.empty((b, k), device="cuda"), torch.empty((b,), dtype=torch.int32, device="cuda"))


ob = outs(64)
ix.search(Q, k, out=ob, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_AUTO)
torch.cuda.synchronize()
o1 = outs(1)
lat, one = [], []
for i in range(12):
    t0 = time.perf_counter()
    ix.search(Q[i:i + 1], k, out=o1, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_AUTO)
    torch.cuda.synchronize()
    lat.append(time.perf_counter() - t0)
    st = ix.stats()
    one.append(int(st["one_pass"]))
    assert torch.equal(o1[0][0], ob[0][i]) and torch.equal(o1[1][0], ob[1][i]), i
print("10M x 768 single-vector calls == the batch's rows; one_pass", one, "p50 ms %.3f" % (1e3 * float(np.median(lat))), "re-ranked rows (last call)", st["rerank_rows"])
