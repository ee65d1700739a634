#!/usr/bin/env python
"""Drives an `epsilla` CPython module (the reference's own build under oracle/_ref/pymod, or the gfx950 drop-in build under
dropin/_build) through the reference's binding API and prints one JSON document.  Run in a fresh process with the
module directory first on sys.path:
    python scripts/epsilla_module_driver.py MODULE_DIR DB_PATH cities|batch|c1 [rows] [dim] [queries]
  cities  the fixture of engine/test/bindings/python/test.py (5 cities, 3 metrics, filter, duplicate PK, delete)
  batch   rows x dim random table: query() one by one, then (drop-in only) rebuild() and query_batch()
  c1      BASELINE configs[0]: rows x dim inserted through insert() in 1000-row JSON batches, `queries` query() calls
  bulk    rows x dim written as the reference's own table files (vectordb_amd/segment_file.py) and loaded by the reference's
          loader (drop-in: load_db_scaled), then query() for the first 16 queries and query_batch() for all of them as ONE NumPy
          matrix, unfiltered and with "ID < rows/2"; args: rows dim queries [metric] [batches]
  ingest  (drop-in only, r6) the SAME records into table J through insert() (JSON, the reference's path) and into table A through
          insert_array() (column buffers): one table with an INT key, INT / FLOAT / DOUBLE / BOOL / STRING attributes and three vector fields
          (EUCLIDEAN, DOT_PRODUCT, COSINE) of `dim` columns; duplicate keys inside and across batches, deletes afterwards; then the same
          query() / query_batch() calls on both, stored (normalised) vectors included in the response
  ingest_big  (drop-in only) rows x dim through insert_array() in chunks of 1M rows into a table loaded with load_db_scaled: seconds, rows/s"""
import json
import sys
import time

sys.path.insert(0, sys.argv[1])
import numpy as np  # noqa: E402
import epsilla  # noqa: E402

db_path, what = sys.argv[2], sys.argv[3]
rows = int(sys.argv[4]) if len(sys.argv) > 4 else 2000
dim = int(sys.argv[5]) if len(sys.argv) > 5 else 32
nq = int(sys.argv[6]) if len(sys.argv) > 6 else 64
out = {"module": getattr(epsilla, "backend", "reference"), "file": epsilla.__file__}


def create_table(name, fields):
    # the reference binding drops a reference it does not own (`Py_DECREF(tableFieldsListPtr)` on a borrowed argument,
    # bindings/python/interface.cpp:129): without this compensation the list is freed twice and the interpreter
    # segfaults at exit - in the reference's own module as well
    import ctypes
    ctypes.pythonapi.Py_IncRef(ctypes.py_object(fields))
    return epsilla.create_table(table_name=name, table_fields=fields)


if what == "bulk":
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from vectordb_amd.segment_file import write_database
    metric = sys.argv[7] if len(sys.argv) > 7 else "EUCLIDEAN"
    batches = int(sys.argv[8]) if len(sys.argv) > 8 else 3
    rng = np.random.default_rng(42)
    t0 = time.perf_counter()
    X = np.empty((rows, dim), np.float32)
    for s in range(0, rows, 1 << 18):
        X[s:s + (1 << 18)] = rng.random((min(1 << 18, rows - s), dim), dtype=np.float32)
    if metric == "COSINE":
        for s in range(0, rows, 1 << 18):
            X[s:s + (1 << 18)] /= np.linalg.norm(X[s:s + (1 << 18)], axis=1, keepdims=True)
    Q = np.random.default_rng(43).random((nq, dim), dtype=np.float32)
    out["generate_s"] = time.perf_counter() - t0
    fields = [{"name": "ID", "dataType": "INT", "primaryKey": True},
              {"name": "V", "dataType": "VECTOR_FLOAT", "dimensions": dim, "metricType": metric}]
    t0 = time.perf_counter()
    write_database(db_path, "T", fields, {"ID": np.arange(rows, dtype=np.int32), "V": X})
    out["write_files_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    if hasattr(epsilla, "load_db_scaled"):
        assert epsilla.load_db_scaled(db_name="db", db_path=db_path, vector_scale=rows + 1024, wal_enabled=False) == 0
    else:
        assert rows <= 150000, "the reference binding loads at most 150000 rows per table"
        assert epsilla.load_db(db_name="db", db_path=db_path) == 0
    epsilla.use_db(db_name="db")
    out["load_db_s"] = time.perf_counter() - t0
    kw = dict(table_name="T", query_field="V", response_fields=["ID"], limit=10, with_distance=True)
    out["single"] = {}
    for flt in ("", "ID < %d" % (rows // 2)):
        t0 = time.perf_counter()
        res = []
        for q in Q[:16]:
            code, resp = epsilla.query(query_vector=q.tolist(), filter=flt, **kw)
            res.append([[r["ID"] for r in resp], [r["@distance"] for r in resp]])
        out["single"][flt] = {"results": res, "first16_s": time.perf_counter() - t0}
    if hasattr(epsilla, "query_batch"):
        out["batch"] = {}
        for flt in ("", "ID < %d" % (rows // 2)):
            code, resp = epsilla.query_batch(query_vectors=Q, filter=flt, **kw)          # (warm: mirrors, filter program)
            t0 = time.perf_counter()
            for _ in range(batches):
                code, resp = epsilla.query_batch(query_vectors=Q, filter=flt, **kw)
            sec = (time.perf_counter() - t0) / batches
            assert code == 0 and len(resp) == nq
            out["batch"][flt] = {"qps": nq / sec, "ms_per_batch": 1e3 * sec, "queries": nq, "batches": batches,
                                 "results": [[[r["ID"] for r in rr], [r["@distance"] for r in rr]] for rr in resp[:64]]}
            # r5: the same batch answered as NumPy arrays (no per-row Python objects)
            code, arr = epsilla.query_batch(query_vectors=Q, filter=flt, as_arrays=True, **kw)
            t0 = time.perf_counter()
            for _ in range(batches):
                code, arr = epsilla.query_batch(query_vectors=Q, filter=flt, as_arrays=True, **kw)
            sec = (time.perf_counter() - t0) / batches
            same = all([r["ID"] for r in resp[i]] == arr["ID"][i][:arr["@count"][i]].tolist() and
                       np.allclose([r["@distance"] for r in resp[i]], arr["@distance"][i][:arr["@count"][i]], rtol=1e-6, atol=0) for i in range(len(resp)))
            out["batch_arrays"] = out.get("batch_arrays", {})
            out["batch_arrays"][flt] = {"qps": nq / sec, "ms_per_batch": 1e3 * sec, "equals_the_dict_form": bool(same), "shapes": [list(arr["ID"].shape), list(arr["@distance"].shape), list(arr["@count"].shape)],
                                        "dtypes": [str(arr["ID"].dtype), str(arr["@distance"].dtype), str(arr["@count"].dtype)]}
        code, resp2 = epsilla.query_batch(query_vectors=[q.tolist() for q in Q[:8]], filter="", **kw)   # list-of-lists form
        out["batch"]["lists"] = [[[r["ID"] for r in rr], [r["@distance"] for r in rr]] for rr in resp2]
        code, resp3 = epsilla.query_batch(query_vectors=Q[:4].astype(np.float64), filter="", table_name="T", query_field="V", response_fields=[],
                                          limit=3, with_distance=False)                               # all fields, float64 queries
        out["batch"]["all_fields_keys"] = sorted(resp3[0][0].keys())
        out["batch"]["all_fields_ids"] = [[r["ID"] for r in rr] for rr in resp3]
        if os.environ.get("EPS_MODULE_REBUILD") and hasattr(epsilla, "rebuild"):
            # the reference's semantics from here on: Rebuild() builds the NSG (on the field's device mirror), searches walk it
            exact64 = out["batch"][""]["results"]
            t0 = time.perf_counter()
            out["rebuild_code"] = epsilla.rebuild()
            out["rebuild_s"] = time.perf_counter() - t0
            code, resp = epsilla.query_batch(query_vectors=Q, filter="", **kw)
            t0 = time.perf_counter()
            for _ in range(batches):
                code, resp = epsilla.query_batch(query_vectors=Q, filter="", **kw)
            sec = (time.perf_counter() - t0) / batches
            assert code == 0 and len(resp) == nq
            hit = sum(len(set(r["ID"] for r in resp[i]) & set(exact64[i][0])) for i in range(len(exact64)))
            code1, resp1 = epsilla.query(query_vector=Q[0].tolist(), filter="", **kw)
            out["graph"] = {"qps": nq / sec, "ms_per_batch": 1e3 * sec, "queries": nq, "batches": batches,
                            "recall_at_10_vs_exact_first_64": hit / float(10 * len(exact64)),
                            "query()_equals_query_batch()[0]": [r["ID"] for r in resp1] == [r["ID"] for r in resp[0]]}
        # exact ground truth of the first queries straight from the rows (numpy, float64)
    gt = []
    for q in Q[:4]:
        d = ((X.astype(np.float64) - q) ** 2).sum(1) if metric == "EUCLIDEAN" and rows <= 2_000_000 else None
        gt.append(None if d is None else np.argsort(d, kind="stable")[:10].tolist())
    out["numpy_top10"] = gt
    print("EPSILLA_JSON " + json.dumps(out))
    sys.exit(0)

if what in ("ingest", "ingest_big"):
    assert hasattr(epsilla, "insert_array"), "the drop-in module only"
    if what == "ingest_big":
        metric = sys.argv[7] if len(sys.argv) > 7 else "COSINE"
        assert epsilla.load_db_scaled(db_name="db", db_path=db_path, vector_scale=rows + 1024, wal_enabled=False) == 0
        epsilla.use_db(db_name="db")
        create_table("T", [{"name": "ID", "dataType": "INT", "primaryKey": True},
                           {"name": "V", "dataType": "VECTOR_FLOAT", "dimensions": dim, "metricType": metric}])
        rng = np.random.default_rng(42)
        chunk = 1 << 20
        buf = np.empty((chunk, dim), np.float32)
        gen_s = ins_s = 0.0
        first = None
        for s in range(0, rows, chunk):
            m = min(chunk, rows - s)
            t0 = time.perf_counter()
            for a in range(0, m, 1 << 17):
                c = min(1 << 17, m - a)
                buf[a:a + c] = rng.random((c, dim), dtype=np.float32)
            if first is None:
                first = buf[:64].copy()
            gen_s += time.perf_counter() - t0
            t0 = time.perf_counter()
            code, r = epsilla.insert_array(table_name="T", columns={"ID": np.arange(s, s + m, dtype=np.int32), "V": buf[:m]})
            ins_s += time.perf_counter() - t0
            assert code == 0 and r == {"inserted": m, "skipped": 0}, (code, r)
        out["rows"], out["dim"], out["metric"] = rows, dim, metric
        out["generate_s"], out["insert_array_s"], out["rows_per_s"] = gen_s, ins_s, rows / ins_s
        kw = dict(table_name="T", query_field="V", response_fields=["ID"], limit=10, with_distance=True)
        t0 = time.perf_counter()
        code, resp = epsilla.query(query_vector=first[0].tolist(), filter="", **kw)       # the first search uploads the table to the device
        out["first_query_s"] = time.perf_counter() - t0
        out["first_query_top"] = [resp[0]["ID"], resp[0]["@distance"]]
        t0 = time.perf_counter()
        code, arr = epsilla.query_batch(query_vectors=first, filter="", as_arrays=True, **kw)
        out["batch64_s"] = time.perf_counter() - t0
        out["batch64_self_hits"] = int((arr["ID"][:, 0] == np.arange(64)).sum())              # every query is a table row: its own nearest neighbour
        print("EPSILLA_JSON " + json.dumps(out))
        sys.exit(0)
    assert epsilla.load_db(db_name="db", db_path=db_path) == 0
    epsilla.use_db(db_name="db")
    fields = [{"name": "ID", "dataType": "INT", "primaryKey": True}, {"name": "Tag", "dataType": "INT"}, {"name": "Score", "dataType": "FLOAT"},
              {"name": "Weight", "dataType": "DOUBLE"}, {"name": "Flag", "dataType": "BOOL"}, {"name": "Name", "dataType": "STRING"},
              {"name": "VL2", "dataType": "VECTOR_FLOAT", "dimensions": dim, "metricType": "EUCLIDEAN"},
              {"name": "VIP", "dataType": "VECTOR_FLOAT", "dimensions": dim, "metricType": "DOT_PRODUCT"},
              {"name": "VCOS", "dataType": "VECTOR_FLOAT", "dimensions": dim, "metricType": "COSINE"}]
    create_table("J", [dict(f) for f in fields])
    create_table("A", [dict(f) for f in fields])
    rng = np.random.default_rng(42)
    X = rng.random((rows, dim), dtype=np.float32) - np.float32(0.25)
    X[7] = 0.0                                                      # a zero vector: COSINE leaves it alone (|v|^2 <= 1e-10)
    X[8] = 1e-7                                                     # ... and one just below the rule's threshold
    ids = np.arange(rows, dtype=np.int64)
    ids[rows // 3] = 5                                              # duplicate keys: inside a batch, ...
    ids[rows - 1] = 17                                              # ... across batches, and as the LAST record of the last batch
    tag = rng.integers(0, 100, rows).astype(np.int32)
    score = rng.random(rows, dtype=np.float32)
    weight = rng.random(rows)
    flag = rng.integers(0, 2, rows).astype(np.bool_)
    names = ["n%d" % (i % 977) for i in range(rows)]
    Q = np.random.default_rng(43).random((nq, dim), dtype=np.float32) - np.float32(0.25)
    step = 1000
    t0 = time.perf_counter()
    codes = []
    for s in range(0, rows, step):
        e = min(rows, s + step)
        codes.append(epsilla.insert(table_name="J", records=[
            {"ID": int(ids[i]), "Tag": int(tag[i]), "Score": float(score[i]), "Weight": float(weight[i]), "Flag": bool(flag[i]), "Name": names[i],
             "VL2": X[i].tolist(), "VIP": X[i].tolist(), "VCOS": X[i].tolist()} for i in range(s, e)]))
    out["json_insert_s"] = time.perf_counter() - t0
    out["json_codes_ok"] = all(c == 0 for c in codes)
    t0 = time.perf_counter()
    res = []
    big = 7 * step                                                   # (different batch boundaries: the segment state must not depend on them)
    for s in range(0, rows, big):
        e = min(rows, s + big)
        res.append(epsilla.insert_array(table_name="A", columns={
            "ID": ids[s:e], "Tag": tag[s:e], "Score": score[s:e], "Weight": weight[s:e], "Flag": flag[s:e], "Name": names[s:e],
            "VL2": X[s:e], "VIP": X[s:e].astype(np.float64), "VCOS": X[s:e]}))
    out["array_insert_s"] = time.perf_counter() - t0
    out["array_results"] = res
    gone = [3, 11, int(ids[rows // 2]), int(ids[rows - 2])]
    out["delete_codes"] = [epsilla.delete(table_name=t, primary_keys=gone) for t in ("J", "A")]
    out["answers"] = {}
    for t in ("J", "A"):
        ans = {}
        for field in ("VL2", "VIP", "VCOS"):
            for flt in ("", "Tag < 50", "Flag = true AND Score < 0.5"):
                rows_out = []
                for q in Q[:8]:
                    code, resp = epsilla.query(table_name=t, query_field=field, response_fields=["ID", "Tag", "Score", "Weight", "Flag", "Name", field],
                                               query_vector=q.tolist(), filter=flt, limit=10, with_distance=True)
                    rows_out.append([code, resp])
                ans["query|%s|%s" % (field, flt)] = rows_out
                code, resp = epsilla.query_batch(table_name=t, query_field=field, query_vectors=Q, response_fields=["ID"], limit=10, filter=flt,
                                                 with_distance=True)
                ans["batch|%s|%s" % (field, flt)] = [code, [[[r["ID"] for r in rr], [r["@distance"] for r in rr]] for rr in resp]]
        out["answers"][t] = ans
    # after rebuild() (graphs built on the device mirror of both tables; the build is deterministic): the graph path answers both tables alike
    t0 = time.perf_counter()
    out["rebuild_code"] = epsilla.rebuild()
    out["rebuild_s"] = time.perf_counter() - t0
    out["answers_after_rebuild"] = {}
    for t in ("J", "A"):
        ans = {}
        for field in ("VL2", "VIP", "VCOS"):
            code, resp = epsilla.query_batch(table_name=t, query_field=field, query_vectors=Q[:16], response_fields=["ID"], limit=10, filter="", with_distance=True)
            ans[field] = [code, [[[r["ID"] for r in rr], [r["@distance"] for r in rr]] for rr in resp]]
        out["answers_after_rebuild"][t] = ans
    # beyond the capacity (the binding loads databases with 150000 rows per table): both paths refuse, with the same text
    over = 150001 - rows + 2                                         # (two records were skipped as duplicates: rows - 2 are stored)
    try:
        epsilla.insert_array(table_name="A", columns={
            "ID": np.arange(10 ** 6, 10 ** 6 + over, dtype=np.int64), "Tag": np.zeros(over, np.int32), "Score": np.zeros(over, np.float32),
            "Weight": np.zeros(over), "Flag": np.zeros(over, np.bool_), "Name": ["x"] * over, "VL2": np.zeros((over, dim), np.float32),
            "VIP": np.zeros((over, dim), np.float32), "VCOS": np.zeros((over, dim), np.float32)})
        out["capacity_error"] = None
    except Exception as e:  # noqa: BLE001
        out["capacity_error"] = str(e)
    try:
        epsilla.insert_array(table_name="A", columns={"ID": ids[:4]})
        out["missing_field_error"] = None
    except Exception as e:  # noqa: BLE001
        out["missing_field_error"] = str(e)
    code, r = epsilla.insert_array(table_name="A", columns={
        "ID": np.array([5, 10 ** 7], np.int64), "Tag": np.array([1, 2], np.int32), "Score": np.zeros(2, np.float32), "Weight": np.zeros(2),
        "Flag": np.ones(2, np.bool_), "Name": ["upserted", "new"], "VL2": Q[:2], "VIP": Q[:2], "VCOS": Q[:2]}, upsert=True)
    out["upsert"] = [code, r]
    code, resp = epsilla.query(table_name="A", query_field="VL2", response_fields=["ID", "Name"], query_vector=Q[0].tolist(), filter="", limit=2, with_distance=True)
    out["after_upsert"] = resp
    print("EPSILLA_JSON " + json.dumps(out))
    sys.exit(0)

assert epsilla.load_db(db_name="db", db_path=db_path) == 0
epsilla.use_db(db_name="db")

if what == "cities":
    create_table("MyTable", [
        {"name": "ID", "dataType": "INT", "primaryKey": True}, {"name": "Doc", "dataType": "STRING"},
        {"name": "EmbeddingEuclidean", "dataType": "VECTOR_FLOAT", "dimensions": 4, "metricType": "EUCLIDEAN"},
        {"name": "EmbeddingDotProduct", "dataType": "VECTOR_FLOAT", "dimensions": 4, "metricType": "DOT_PRODUCT"},
        {"name": "EmbeddingCosine", "dataType": "VECTOR_FLOAT", "dimensions": 4, "metricType": "COSINE"}])
    cities = [(1, "Berlin", [0.05, 0.61, 0.76, 0.74]), (2, "London", [0.19, 0.81, 0.75, 0.11]), (3, "Moscow", [0.36, 0.55, 0.47, 0.94]),
              (4, "San Francisco", [0.18, 0.01, 0.85, 0.80]), (5, "Shanghai", [0.24, 0.18, 0.22, 0.44]), (1, "Berlin", [0.05, 0.61, 0.76, 0.74])]
    epsilla.insert(table_name="MyTable", records=[{"ID": i, "Doc": c, "EmbeddingEuclidean": v, "EmbeddingDotProduct": v, "EmbeddingCosine": v}
                                                  for i, c, v in cities])
    out["queries"] = {}
    for field in ["EmbeddingEuclidean", "EmbeddingDotProduct", "EmbeddingCosine"]:
        for flt in ("ID < 6", "", "ID >= 3", "Doc = 'Moscow' OR ID = 5", "NOT (ID < 3)"):
            code, resp = epsilla.query(table_name="MyTable", query_field=field, response_fields=["ID", "Doc", field],
                                       query_vector=[0.35, 0.55, 0.47, 0.94], filter=flt, limit=6, with_distance=True)
            out["queries"]["%s|%s" % (field, flt)] = [code, resp]
    out["delete"] = epsilla.delete(table_name="MyTable", primary_keys=[1, 2, 3, 4])
    out["after_delete"] = epsilla.query(table_name="MyTable", query_field="EmbeddingEuclidean", response_fields=["ID", "Doc", "EmbeddingEuclidean"],
                                        query_vector=[0.35, 0.55, 0.47, 0.94], filter="ID < 6", limit=10, with_distance=True)
    out["drop"] = epsilla.drop_table("MyTable")
else:
    rng = np.random.default_rng(42)
    X = rng.random((rows, dim), dtype=np.float32)
    Q = np.random.default_rng(43).random((nq, dim), dtype=np.float32)
    create_table("T", [{"name": "ID", "dataType": "INT", "primaryKey": True},
                       {"name": "V", "dataType": "VECTOR_FLOAT", "dimensions": dim, "metricType": "EUCLIDEAN"}])
    t0 = time.perf_counter()
    for s in range(0, rows, 1000):
        assert epsilla.insert(table_name="T", records=[{"ID": int(i), "V": X[i].tolist()} for i in range(s, min(rows, s + 1000))]) == 0
    out["insert_s"] = time.perf_counter() - t0

    def one_by_one():
        lat, res = [], []
        t0 = time.perf_counter()
        for q in Q:
            t1 = time.perf_counter()
            code, resp = epsilla.query(table_name="T", query_field="V", response_fields=["ID"], query_vector=q.tolist(), filter="", limit=10,
                                       with_distance=True)
            lat.append(time.perf_counter() - t1)
            res.append([[r["ID"] for r in resp], [r["@distance"] for r in resp]])
        return time.perf_counter() - t0, lat, res

    one_by_one() if what == "c1" and nq <= 16 else None
    sec, lat, res = one_by_one()
    out["flat"] = {"qps": nq / sec, "p50_ms": 1e3 * float(np.median(lat)), "p99_ms": 1e3 * float(np.percentile(lat, 99)), "results": res}
    if hasattr(epsilla, "rebuild"):
        t0 = time.perf_counter()
        out["rebuild_code"] = epsilla.rebuild()
        out["rebuild_s"] = time.perf_counter() - t0
        sec, lat, res = one_by_one()
        out["graph"] = {"qps": nq / sec, "p50_ms": 1e3 * float(np.median(lat)), "results": res}
        t0 = time.perf_counter()
        code, resp = epsilla.query_batch(table_name="T", query_field="V", query_vectors=[q.tolist() for q in Q], response_fields=["ID"], limit=10,
                                         filter="", with_distance=True)
        sec = time.perf_counter() - t0
        out["query_batch"] = {"code": code, "qps": nq / sec, "results": [[[r["ID"] for r in rr], [r["@distance"] for r in rr]] for rr in resp]}
    epsilla.drop_table("T")
print("EPSILLA_JSON " + json.dumps(out))
