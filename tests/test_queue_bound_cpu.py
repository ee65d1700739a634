"""CPU: the occupancy bound the device traversal lays its worker queues out with (csrc/traverse.hip, r6).

Between two MergeAllQueuesToMaster calls (engine/db/execution/vec_search_executor.cpp:297-326, which empty the worker queues) a worker's queue holds
its share of the master's unchecked candidates - PickTopMToWorkers (:328-356) deals them round-robin, every T-th stays with the master: at most
ceil(L / T) - plus what its <= GlobalSyncInterval expansions insert, at most the out-degree each (ExpandOneCandidate, :384-444).  A LocalQueueSize
above that bound never clamps an insert and never stops the dealing early, so the walk with the smaller capacity must be the walk with the larger
one, key for key.  Checked here on the oracle (bit-exact against the compiled reference, tests/test_oracle_vs_ref.py) under the deterministic lockstep
schedule the device implements and under the oracle's other schedule (one worker after the other); the device side of the same statement is
tests/test_gpu_traverse.py::test_lockstep_workers_match_oracle, where the oracle keeps the caller's LocalQueueSize and the kernel uses the bound."""
import os

import numpy as np
import pytest

from helpers import data

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _graph():
    z = np.load(os.path.join(G, "graph2000x32.npz"))
    off, nbr = z["off"].astype(np.int64), z["nbr"].astype(np.int64)
    return off, nbr, int(z["nav"]), int(np.diff(off).max())


@pytest.mark.parametrize("T,L,I", [(2, 500, 15), (4, 500, 15), (4, 500, 1), (4, 500, 3), (4, 1000, 4), (8, 300, 2), (16, 500, 1), (32, 500, 15), (3, 64, 2), (4, 2000, 15)])
@pytest.mark.parametrize("lockstep", [True, False])
def test_worker_queue_capacity_above_the_occupancy_bound_changes_nothing(oracle, T, L, I, lockstep):
    off, nbr, nav, maxdeg = _graph()
    X, Q = data(2000, 32, 42), data(6, 32, 48)
    L = min(L, 2000)
    bound = (L + T - 1) // T + I * maxdeg
    init = oracle.prepare_init_ids(off, nbr, nav, L)
    for q in Q:
        full = oracle.search_impl(0, X, off, nbr, init, q, T=T, L=L, Lq=L, I=I, lockstep=lockstep)
        cut = oracle.search_impl(0, X, off, nbr, init, q, T=T, L=L, Lq=min(L, bound), I=I, lockstep=lockstep)
        assert np.array_equal(full[0], cut[0]) and np.array_equal(full[1].view(np.uint32), cut[1].view(np.uint32)) and full[2] == cut[2], (T, L, I, bound)
