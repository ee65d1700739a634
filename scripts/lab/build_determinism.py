import os
import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
os.environ.setdefault("EPS_TUNING_FROM_ENV", "1")   # (scripts steer the library's engine switches through the environment: vectordb_amd/_lib.py)
import vectordb_amd as amd
from helpers import data
for n, d in ((60000, 256), (80000, 32), (80000, 256)):
    X = data(n, d, 23)
    gs = []
    for _ in range(3):
        ix = amd.GpuIndex(d, 0); ix.attach_rows(X); ix.build(); gs.append(ix.get_graph()); ix.close()
    same = [bool(np.array_equal(gs[0][0], g[0]) and np.array_equal(gs[0][1], g[1])) for g in gs[1:]]
    diff = [int((np.diff(gs[0][0]) != np.diff(g[0])).sum()) for g in gs[1:]]
    print(n, d, 'same', same, 'nodes with different degree', diff, 'edges', [int(g[0][-1]) for g in gs], 'nav', [g[2] for g in gs])
