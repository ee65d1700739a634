#!/usr/bin/env python
"""What a cluster-restricted kNN stage could find ("cluster-restricted GEMM"): k-means the table, let every
row look for its 100 nearest neighbours only inside the P clusters nearest to it, and count how many of its TRUE 100 nearest
neighbours (exact scan) live there - against the share of the table those clusters hold (= the share of the quadratic work).
    python scripts/lab/ivf_recall_probe.py [rows] [dim] [uniform|manifold] [clusters]      -> JSON lines"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("EPS_TUNING_FROM_ENV", "1")   # (scripts steer the library's engine switches through the environment: vectordb_amd/_lib.py)
import vectordb_amd as amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
kind = sys.argv[3] if len(sys.argv) > 3 else "uniform"
C = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
g = torch.Generator(device="cuda").manual_seed(42)
if kind == "manifold":
    A = 0.25 * torch.randn((16, d), generator=torch.Generator(device="cuda").manual_seed(41), device="cuda")
    X = torch.rand((n, 16), generator=g, device="cuda") @ A + 0.01 * torch.randn((n, d), generator=g, device="cuda")
else:
    X = torch.rand((n, d), generator=g, device="cuda")
nq, K = 1024, 100
Q = X[:nq].contiguous()
ix = amd.GpuIndex(d, 0).use_torch_stream()
ix.attach_rows(X)
ids = torch.empty((nq, K + 1), dtype=torch.int64, device="cuda")
dist = torch.empty((nq, K + 1), device="cuda")
cnt = torch.empty((nq,), dtype=torch.int32, device="cuda")
ix.search(Q, K + 1, out=(ids, dist, cnt), mode=amd.MODE_FLAT)
torch.cuda.synchronize()
gt = ids[:, 1:]   # (position 0 is the row itself)

# Lloyd on a sample, then assign everything
S = X[torch.randperm(n, generator=g, device="cuda")[: min(n, 200_000)]]
cent = S[:C].clone()


def assign(Y, cent):
    out = torch.empty((Y.shape[0],), dtype=torch.int64, device="cuda")
    c2 = (cent * cent).sum(1)
    for s in range(0, Y.shape[0], 65536):
        y = Y[s:s + 65536]
        out[s:s + 65536] = (c2[None, :] - 2.0 * (y @ cent.T)).argmin(1)
    return out


for it in range(8):
    a = assign(S, cent)
    sums = torch.zeros_like(cent).index_add_(0, a, S)
    cnts = torch.bincount(a, minlength=C).clamp(min=1).float()
    cent = sums / cnts[:, None]
lab = assign(X, cent)
sizes = torch.bincount(lab, minlength=C).float()
qc = ((cent * cent).sum(1)[None, :] - 2.0 * (Q @ cent.T)).argsort(1)   # clusters of every query, nearest first
gl = lab[gt]                                                             # [nq][K] cluster of every true neighbour
for P in (1, 4, 16, 64, 256, C):
    P = min(P, C)
    probed = torch.zeros((nq, C), dtype=torch.bool, device="cuda")
    probed.scatter_(1, qc[:, :P], True)
    found = probed.gather(1, gl).float().mean().item()
    share = (probed.float() @ sizes).mean().item() / n
    print(json.dumps({"data": "%s %d x %d" % (kind, n, d), "clusters": C, "probed": P, "share_of_table_scanned": share,
                      "true_100nn_inside": found}), flush=True)
