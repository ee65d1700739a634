"""r6: the one-pass search against the fp32 stream engine, call by call, on the three kinds of tables it now serves - homogeneous rows (table-wide margin),
rows with outliers (margins folded per call; a forced row), embedding-like rows in the rotated frame (cut grid, folded margins) - for random query counts (1..32),
k, metrics, with and without a deleted bitset.  Prints the calls made, how many the one-pass form answered, and any call whose answer differs (none may)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["EPS_TUNING_FROM_ENV"] = "1"
import vectordb_amd as amd  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(2026)
bad = calls = one = 0
for kind in ("uniform", "outliers", "embedding"):
    for metric in (0, 1, 2):
        d = int(rng.choice([512, 768, 1000]))
        n = int(rng.integers(70_000, 200_000))
        if kind == "embedding":
            X = rng.standard_normal((n, d), dtype=np.float32)
            X[:, :8] *= 4.0
            X /= np.linalg.norm(X, axis=1, keepdims=True)
            Q = rng.standard_normal((32, d), dtype=np.float32)
            Q[:, :8] *= 4.0
            Q /= np.linalg.norm(Q, axis=1, keepdims=True)
        else:
            X = rng.random((n, d), dtype=np.float32)
            Q = rng.random((32, d), dtype=np.float32)
            if kind == "outliers":
                X[123, 5] = 90.0
                X[n // 2, 7] = -35.0
                if metric == 0:
                    X[n - 9, 11] = 25_000.0
                Q[3] = X[123]
        X[4000:4030] = X[3999]                      # a run of identical rows
        ix = amd.GpuIndex(d, metric)
        ix.attach_rows(X)
        gone = np.zeros((n + 7) // 8, np.uint8)
        for r in range(5, n, 13):
            gone[r >> 3] |= 1 << (r & 7)
        for it in range(rounds):
            nq = int(rng.choice([1, 1, 1, 2, 3, 4, 5, 8, 13, 16, 17, 24, 32]))
            k = int(rng.choice([1, 10, 10, 16, 17, 40, 64]))
            ix.set_deleted(gone if it % 3 == 2 else None)
            qs = Q[rng.permutation(32)[:nq]]
            a = ix.search(qs, k, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
            st = ix.stats()
            b = ix.search(qs, k, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
            calls += 1
            one += st["one_pass"]
            if not (np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and np.array_equal(a[2], b[2])):
                bad += 1
                print("DIFFERENT:", kind, "metric", metric, "n", n, "d", d, "nq", nq, "k", k, {x: st[x] for x in ("one_pass", "i8_folded", "i8_rotated", "main_kernel_bits")}, flush=True)
        print(kind, "metric", metric, "n", n, "d", d, "folded", st["i8_folded"], "rotated", st["i8_rotated"], "calls so far", calls, "one-pass", one, "different", bad, flush=True)
        ix.close()
print("calls", calls, "answered by the one-pass form", one, "different answers", bad)
sys.exit(1 if bad else 0)
