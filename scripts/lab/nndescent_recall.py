#!/usr/bin/env python
"""Recall of the REFERENCE's NN-Descent lists (knn.hpp:90-135 / nndescent.hpp:96-192, compiled verbatim in oracle/_ref, its
defaults) against exact kNN, on this machine's CPU - the data behind DESIGN.md 0.1 item 5.  Lab only (uses the checker).
    python scripts/lab/nndescent_recall.py rows dim [threads]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.pyoracle import Ref  # noqa: E402

n, d = int(sys.argv[1]), int(sys.argv[2])
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 32
K = 100
X = np.random.default_rng(42).random((n, d), dtype=np.float32)
ref = Ref()
t0 = time.perf_counter()
knn = ref.knn_graph(X, K=K, metric=0, threads=threads)
dt = time.perf_counter() - t0
sample = np.random.default_rng(1).choice(n, size=min(n, 500), replace=False)
hit = 0
x2 = (X * X).sum(1)
for v in sample:
    dist = x2 - 2.0 * (X @ X[v])
    dist[v] = np.inf
    truth = set(np.argpartition(dist, K)[:K].tolist())
    hit += len(truth & set(int(u) for u in knn[v] if u >= 0))
print({"rows": n, "dim": d, "threads": threads, "nndescent_s": round(dt, 2), "recall_at_100": hit / float(K * len(sample))})
