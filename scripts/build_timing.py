"""Build timing + recall of the default search on the built graph:  python scripts/build_timing.py rows dim [data=uniform|manifold]
(EPS_DEBUG=1 prints the stages; EPS_BUILD_BITMAP=1 switches the Link searches to exact visited bitmaps)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("EPS_TUNING_FROM_ENV", "1")   # (scripts steer the library's engine switches through the environment: vectordb_amd/_lib.py)
import vectordb_amd as amd  # noqa: E402

n, d = int(sys.argv[1]), int(sys.argv[2])
kind = sys.argv[3] if len(sys.argv) > 3 else "uniform"
g = torch.Generator(device="cuda").manual_seed(42)
if kind == "manifold":
    A = 0.25 * torch.randn((16, d), generator=torch.Generator(device="cuda").manual_seed(41), device="cuda")
    X = torch.rand((n, 16), generator=g, device="cuda") @ A + 0.01 * torch.randn((n, d), generator=g, device="cuda")
    Q = torch.rand((1024, 16), generator=g, device="cuda") @ A + 0.01 * torch.randn((1024, d), generator=g, device="cuda")
else:
    X = torch.rand((n, d), generator=g, device="cuda")
    Q = torch.rand((1024, d), generator=g, device="cuda")
ix = amd.GpuIndex(d, 0).use_torch_stream()
ix.attach_rows(X)
torch.cuda.synchronize()
t0 = time.perf_counter()
ix.build()
torch.cuda.synchronize()
print("build_s", time.perf_counter() - t0, ix.graph_info())


def outs():
    return (torch.empty((1024, 10), dtype=torch.int64, device="cuda"), torch.empty((1024, 10), device="cuda"), torch.empty((1024,), dtype=torch.int32, device="cuda"))


gt, res = outs(), outs()
ix.search(Q, 10, out=gt, mode=amd.MODE_FLAT)
for L in (100, 500):
    ix.search(Q, 10, out=res, mode=amd.MODE_GRAPH, intra_threads=4, master_queue=L, local_queue=L)
    torch.cuda.synchronize()
    a, b = res[0].cpu().numpy(), gt[0].cpu().numpy()
    print("recall@10 T=4 L=%d: %.4f  evals/query %.0f" % (L, float(np.mean([len(set(x) & set(y)) / 10.0 for x, y in zip(a, b)])), ix.stats()["dist_evals"] / 1024.0))
