"""Concurrent single-vector clients against the drop-in DBServer (reference DBMS layers + gfx950 executor):
throughput with the adapter's micro-batcher on vs. off (EPS_DROPIN_BATCH=0), and the reference's CPU executor beside it.
Usage: python scripts/bench_dropin_mt.py [n] [dim] [nq] [threads]"""
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def run(which, n, d, nq, threads):
    from oracle.pyoracle import DROPIN_SO, Ref
    lib = Ref(DROPIN_SO) if which == "dropin" else Ref()
    lib.L.ref_config(4, 500, 1, 0, 16)
    db = lib.db(os.path.join(tempfile.mkdtemp(), "db"), scale=n + 1000, wal=False)
    rng = np.random.default_rng(5)
    X = rng.standard_normal((n, d), dtype=np.float32)
    schema = {"name": "T", "fields": [{"name": "ID", "dataType": "INT", "primaryKey": True},
                                       {"name": "V", "dataType": "VECTOR_FLOAT", "dimensions": d, "metricType": "EUCLIDEAN"}]}
    assert db.create_table(schema) == 0
    t0 = time.time()
    for s in range(0, n, 20000):
        assert db.insert("T", [{"ID": int(i), "V": X[i].tolist()} for i in range(s, min(n, s + 20000))]) == 0
    Q = rng.standard_normal((nq, d), dtype=np.float32)
    db.search_mt("T", "V", Q[:threads], 10, threads)          # warm-up (row upload)
    sec, first = db.search_mt("T", "V", Q, 10, threads)
    print("%s batch=%s n=%d d=%d threads=%d: %d queries in %.3f s = %.0f QPS (ingest %.1f s) checksum %d" %
          (which, os.environ.get("EPS_DROPIN_BATCH", "1"), n, d, threads, nq, sec, nq / sec, time.time() - t0, int(first.sum())), flush=True)
    db.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] in ("dropin", "ref"):
        run(sys.argv[1], *[int(x) for x in sys.argv[2:6]])
    else:
        n, d, nq, th = [int(x) for x in (sys.argv[1:5] + ["200000", "128", "4096", "64"][len(sys.argv) - 1:])]
        for which, env in (("dropin", "1"), ("dropin", "0"), ("ref", "1")):
            e = dict(os.environ, EPS_DROPIN_BATCH=env)
            subprocess.run([sys.executable, __file__, which, str(n), str(d), str(nq if which == "dropin" else max(64, nq // 16)), str(th)], env=e, check=False)
