#!/usr/bin/env python
"""Generates tests/golden/*.npz from the REFERENCE ITSELF (oracle/_ref/libepsilla_ref.so = the reference's
own sources compiled verbatim, see oracle/Makefile).  Run in the build container, where /root/reference
exists:   python scripts/gen_golden.py
Inputs are regenerated from seeds by the tests, so only reference OUTPUTS are stored."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Oracle, Ref, build_oracle, build_ref  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def data(n, d, seed):
    return np.random.default_rng(seed).random((n, d), dtype=np.float32)


def main():
    build_oracle()
    build_ref()
    ref, orc = Ref(), Oracle()
    os.makedirs(OUT, exist_ok=True)

    # 1. distances (fvec_L2sqr / fvec_inner_product + the three metric epilogues)
    dd = {}
    for d in (1, 3, 4, 7, 33, 128, 768):
        X, Q = data(32, d, 100 + d), data(3, d, 200 + d) * 2 - 0.5
        for m in (0, 1, 2):
            dd["d%d_m%d" % (d, m)] = np.stack([ref.dist_batch(m, X, q) for q in Q])
    np.savez_compressed(os.path.join(OUT, "distances.npz"), **dd)

    # 2. reference-built graph (NN-Descent -> NSG, 1 thread) + T=1 SearchImpl master queues
    X = data(2000, 32, 42)
    g = ref.build_graph(X, metric=0, threads=1)
    off, nbr, nav = ref.graph_arrays(g)
    Q = data(16, 32, 43)
    out = {"off": off.astype(np.int32), "nbr": nbr.astype(np.int32), "nav": np.int64(nav)}
    for m in (0, 1, 2):
        ex = ref.executor(g, X, metric=m, T=1, L=500)
        out["init"] = ref.init_ids(ex, 500).astype(np.int32)
        res = [ref.search_impl(ex, q, 500) for q in Q]
        out["ids_m%d" % m] = np.stack([r[0] for r in res]).astype(np.int32)
        out["dist_m%d" % m] = np.stack([r[1] for r in res])
    np.savez_compressed(os.path.join(OUT, "graph2000x32.npz"), **out)

    # 3. NsgIndex::Build on an exact kNN graph (the kNN input is recomputed by the test)
    X = data(600, 16, 7)
    knn = orc.knn_exact(0, X, 100)
    g = ref.nsg_from_knn(X, knn, threads=1, seed=100)
    off, nbr, nav = ref.graph_arrays(g)
    np.savez_compressed(os.path.join(OUT, "nsg600x16.npz"), off=off.astype(np.int32), nbr=nbr.astype(np.int32),
                        nav=np.int64(nav))

    # 4. DBServer level: graph on the first 1000 rows + brute-force tail of 500, int filter, deletes
    ref.L.ref_config(1, 500, 1, 0, 1)
    tmp = tempfile.mkdtemp()
    db = ref.db(os.path.join(tmp, "db"))
    schema = {"name": "T", "fields": [{"name": "ID", "dataType": "INT", "primaryKey": True},
                                       {"name": "V", "dataType": "VECTOR_FLOAT", "dimensions": 8,
                                        "metricType": "EUCLIDEAN"}]}
    assert db.create_table(schema) == 0
    X = data(1500, 8, 15)
    recs = [{"ID": int(i), "V": [float(x) for x in X[i]]} for i in range(1500)]
    assert db.insert("T", recs[:1000]) == 0 and db.rebuild() == 0 and db.insert("T", recs[1000:]) == 0
    off, nbr, nav, _ = orc.graph_read(os.path.join(tmp, "db", "0", "ann_graph_1.bin"))
    Q = data(8, 8, 16)
    out = {"off": off.astype(np.int32), "nbr": nbr.astype(np.int32), "nav": np.int64(nav)}
    for name, flt in (("plain", ""), ("lt700", "ID < 700"), ("ge1200", "ID >= 1200")):
        for limit in (10, 100):
            ids, ds = [], []
            for q in Q:
                rc, res = db.search("T", "V", q, limit, flt=flt)
                assert rc == 0
                i = [r["ID"] for r in res] + [-1] * (limit - len(res))
                x = [r["@distance"] for r in res] + [np.nan] * (limit - len(res))
                ids.append(i)
                ds.append(x)
            out["%s_k%d_ids" % (name, limit)] = np.array(ids, np.int32)
            out["%s_k%d_dist" % (name, limit)] = np.array(ds, np.float64)
    assert db.delete("T", list(range(0, 1500, 3))) == 0
    ids, ds = [], []
    for q in Q:
        rc, res = db.search("T", "V", q, 10)
        ids.append([r["ID"] for r in res] + [-1] * (10 - len(res)))
        ds.append([r["@distance"] for r in res] + [np.nan] * (10 - len(res)))
    out["deleted3_k10_ids"] = np.array(ids, np.int32)
    out["deleted3_k10_dist"] = np.array(ds, np.float64)
    np.savez_compressed(os.path.join(OUT, "dbserver1500x8.npz"), **out)
    db.close()
    ref.L.ref_config(4, 500, 1, 0, 16)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
