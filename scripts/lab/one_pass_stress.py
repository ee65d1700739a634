"""How often does the one-pass search fall back to the staged chain?  python scripts/lab/one_pass_stress.py [rows] [dim] [calls]   (EPS_DEBUG_ONE_PASS_OVERFLOW=1 prints why)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("EPS_TUNING_FROM_ENV", "1")   # (scripts steer the library's engine switches through the environment: vectordb_amd/_lib.py)
import vectordb_amd as amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 90_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
g = torch.Generator(device="cuda").manual_seed(7)
X = torch.rand((n, d), generator=g, device="cuda")
Q = torch.rand((256, d), generator=g, device="cuda")
ix = amd.GpuIndex(d, 0).use_torch_stream()
ix.attach_rows(X)
def outs(b, k):
    return (torch.empty((b, k), dtype=torch.int64, device="cuda"), torch.empty((b, k), device="cuda"), torch.empty((b,), dtype=torch.int32, device="cuda"))


ix.search(Q[:64], 10, out=outs(64, 10), mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
for nq, k in ((1, 10), (4, 16), (2, 1), (3, 16)):
    fell = 0
    o = outs(nq, k)
    for i in range(calls):
        ix.search(Q[(i * nq) % 250:(i * nq) % 250 + nq], k, out=o, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
        fell += ix.stats()["one_pass"] == 0
    print("rows %d dim %d nq %d k %d: %d of %d calls fell back to the chain" % (n, d, nq, k, fell, calls))
