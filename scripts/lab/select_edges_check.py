import os
"""device SelectEdge (prune_kernel) vs the oracle on identical pools at several dimensions, repeated (race hunting)"""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
os.environ.setdefault("EPS_TUNING_FROM_ENV", "1")   # (scripts steer the library's engine switches through the environment: vectordb_amd/_lib.py)
import vectordb_amd as amd
from helpers import data
from oracle.pyoracle import Oracle
o = Oracle()
for d in (32, 128, 256, 768):
    n = 4000
    X = data(n, d, 17 + d)
    ix = amd.GpuIndex(d, 0); ix.attach_rows(X)
    rng = np.random.default_rng(18)
    nodes = np.arange(0, 900, 3, dtype=np.int64)
    cands = np.stack([rng.choice(n, size=300, replace=False) for _ in nodes]).astype(np.int64)
    for i, v in enumerate(nodes):
        cands[i][cands[i] == v] = -1
    cands[:, 0] = nodes
    want = [o.select_edge(X, int(v), cands[i], 300, 50) for i, v in enumerate(nodes)]
    for rep in range(4):
        ids, deg = ix.select_edges(nodes, cands, depth=300, out_degree=50)
        bad = [int(v) for i, v in enumerate(nodes) if list(ids[i][:deg[i]]) != list(want[i])]
        print('d', d, 'rep', rep, 'mismatching nodes', len(bad), bad[:5])
    ix.close()
