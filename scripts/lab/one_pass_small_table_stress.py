import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["EPS_TUNING_FROM_ENV"] = "1"
os.environ["EPS_DEBUG_ONE_PASS_OVERFLOW"] = "1"
import vectordb_amd as amd
sys.path.insert(0, "tests")
from test_gpu_mfma_i8 import data, bitset
# recycled device memory full of garbage, as in a long test process
junk = [torch.full((1 << 28,), 0x7F, dtype=torch.uint8, device="cuda") for _ in range(8)]
torch.cuda.synchronize()
del junk
torch.cuda.empty_cache()
n, d = 90_000, 768
X, Q = data(n, d, 171 + d), data(16, d, 172 + d)
X[5000:5040] = X[4999]
Q[1] = X[5010]
idc = np.arange(n, dtype=np.int32)
ix = amd.GpuIndex(d, 0)
ix.attach_rows(X)
ix.set_deleted(bitset(n, range(3, n, 11)))
ix.set_int_filter(idc, ">=", 1000)
bad = 0
calls = 0
for loop in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    for nq in (1, 2, 3, 4, 5, 8, 13, 16, 1):
        for k in (1, 10, 16, 17, 40, 64, 10):
            for rep in range(2):
                ix.search(Q[:nq], k, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
                st = ix.stats()
                calls += 1
                if st["one_pass"] != 1:
                    bad += 1
                    print("NOT ONE PASS: loop", loop, "nq", nq, "k", k, "rep", rep, {x: st[x] for x in ("one_pass", "rerank_rows", "overflow_queries", "dist_evals")}, flush=True)
                if loop % 2 == 0:
                    ix.search(Q[:nq], k, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
print("calls", calls, "not one pass", bad)
