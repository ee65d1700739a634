"""The `epsilla` CPython module (SURVEY 8b "Python-module ABI").  oracle/_ref/pymod/epsilla.so is the reference's own binding
over the reference engine; dropin/_build/epsilla.so is the SAME binding source (bindings/python/interface.cpp, unmodified)
over this repository's executor + libepsilla_gfx950.so, plus the additive rebuild() / query_batch().  Each module runs in its
own process (both are called `epsilla`)."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref", "pymod")
GPU_DIR = os.path.join(ROOT, "dropin", "_build")
DRIVER = os.path.join(ROOT, "scripts", "epsilla_module_driver.py")
GOLDEN = os.path.join(ROOT, "tests", "golden", "epsilla_module_cities.json")


def _run(module_dir, what, *args):
    db = os.path.join(tempfile.mkdtemp(), "db")
    r = subprocess.run([sys.executable, DRIVER, module_dir, db, what] + [str(a) for a in args], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("EPSILLA_JSON ")][-1]
    return json.loads(line[len("EPSILLA_JSON "):])


def _have(d):
    return os.path.exists(os.path.join(d, "epsilla.so"))


def _same_rows(a, b, rtol=1e-4):
    assert a[0] == b[0] == 0
    assert [r["ID"] for r in a[1]] == [r["ID"] for r in b[1]], (a, b)
    assert [r["Doc"] for r in a[1]] == [r["Doc"] for r in b[1]]
    assert np.allclose([r["@distance"] for r in a[1]], [r["@distance"] for r in b[1]], rtol=rtol, atol=1e-7)


@pytest.mark.ref
def test_reference_module_reproduces_the_survey_anchors():
    """the reference's own module on the fixture of engine/test/bindings/python/test.py: squared L2, negative dot, 1 - dot on
    stored-normalised rows; after delete([1,2,3,4]) only Shanghai is left.  Also refreshes nothing: the committed golden must
    equal what the reference returns here."""
    if not _have(REF_DIR):
        pytest.skip("oracle/_ref/pymod/epsilla.so not built (make -C oracle pymod)")
    out = _run(REF_DIR, "cities")
    q = out["queries"]
    top = q["EmbeddingEuclidean|ID < 6"][1][0]
    assert top["Doc"] == "Moscow" and abs(top["@distance"] - 1.0000040e-4) < 1e-9
    assert abs(q["EmbeddingDotProduct|ID < 6"][1][0]["@distance"] + 1.5329999924) < 1e-6
    assert abs(q["EmbeddingCosine|ID < 6"][1][0]["@distance"] - 2.99e-5) < 1e-6
    assert [r["Doc"] for r in out["after_delete"][1]] == ["Shanghai"] and abs(out["after_delete"][1][0]["@distance"] - 0.46149999) < 1e-6
    gold = json.load(open(GOLDEN))
    for key in gold["queries"]:
        _same_rows(q[key], gold["queries"][key], rtol=1e-6)


@pytest.mark.gpu
def test_gfx950_module_answers_like_the_reference_module():
    """same binding source over the MI355X executor: identical rows / order / distances for every metric and filter of the
    fixture (golden = the reference module's answers), incl. string and OR / NOT filters and the delete."""
    if not _have(GPU_DIR):
        pytest.skip("dropin/_build/epsilla.so not built (make -C dropin)")
    out = _run(GPU_DIR, "cities")
    assert out["module"] == "gfx950"
    gold = json.load(open(GOLDEN))
    for key in gold["queries"]:
        _same_rows(out["queries"][key], gold["queries"][key])
    _same_rows(out["after_delete"], gold["after_delete"])
    assert out["delete"] == gold["delete"] == 0
    if _have(REF_DIR):   # side by side on this box as well
        ref = _run(REF_DIR, "cities")
        for key in ref["queries"]:
            _same_rows(out["queries"][key], ref["queries"][key])


@pytest.mark.gpu
def test_gfx950_module_rebuild_and_query_batch():
    """additive entry points: rebuild() builds the graph on the device through the unchanged DBServer::Rebuild; query_batch()
    returns what N query() calls return.  3000 x 32: the flat answers equal the reference module's, the graph answers (T = 4,
    L = 500 on 3000 rows: everything is evaluated) equal the flat ones."""
    if not _have(GPU_DIR):
        pytest.skip("dropin/_build/epsilla.so not built (make -C dropin)")
    out = _run(GPU_DIR, "batch", 3000, 32, 48)
    assert out["rebuild_code"] == 0 and out["query_batch"]["code"] == 0
    flat, graph, qb = out["flat"]["results"], out["graph"]["results"], out["query_batch"]["results"]
    for i in range(len(flat)):
        assert flat[i][0] == graph[i][0] == qb[i][0], i
        assert np.allclose(flat[i][1], qb[i][1], rtol=1e-5)
    if _have(REF_DIR):
        ref = _run(REF_DIR, "batch", 3000, 32, 48)
        for i in range(len(flat)):
            assert flat[i][0] == ref["flat"]["results"][i][0], i
            assert np.allclose(flat[i][1], ref["flat"]["results"][i][1], rtol=1e-4)


@pytest.mark.gpu
def test_gfx950_module_query_batch_is_one_device_batch_and_equals_single_queries(tmp_path):
    """query_batch() takes the N vectors as ONE matrix (NumPy float32; also float64 and lists of lists), issues
    ONE eps_index_search(nq = N) through VecSearchExecutor::SearchBatch and projects once.  Element q equals query(vector q) - ids
    position by position, distances to 1e-6 - unfiltered and with an attribute filter (compiled to the device program); the table
    is brought in through the reference's own file loader from files written in its format (100 000 rows: within the reference
    binding's own 150 000-row capacity, so its module loads the same files side by side)."""
    if not _have(GPU_DIR):
        pytest.skip("dropin/_build/epsilla.so not built (make -C dropin)")
    rows, dim, nq = 100_000, 64, 300
    out = _run(GPU_DIR, "bulk", rows, dim, nq)
    for flt in ("", "ID < %d" % (rows // 2)):
        single, batch = out["single"][flt]["results"], out["batch"][flt]["results"]
        for i in range(16):
            assert single[i][0] == batch[i][0], (flt, i, single[i], batch[i])
            assert np.allclose(single[i][1], batch[i][1], rtol=1e-6, atol=0)
            if flt:
                assert max(batch[i][0]) < rows // 2
    for i in range(4):
        assert out["batch"][""]["results"][i][0] == out["numpy_top10"][i], i
    assert out["batch"]["lists"] == out["batch"][""]["results"][:8]
    for flt in ("", "ID < %d" % (rows // 2)):    # r5: as_arrays=True returns the same answer as NumPy arrays
        a = out["batch_arrays"][flt]
        assert a["equals_the_dict_form"] and a["shapes"] == [[nq, 10], [nq, 10], [nq]] and a["dtypes"] == ["int64", "float32", "int32"], a
    assert out["batch"]["all_fields_keys"] == ["ID", "V"]
    assert out["batch"]["all_fields_ids"] == [r[0][:3] for r in out["batch"][""]["results"][:4]]
    if _have(REF_DIR):   # the reference's own module over the same files
        ref = _run(REF_DIR, "bulk", rows, dim, 16)
        for flt in ("", "ID < %d" % (rows // 2)):
            for i in range(16):
                assert ref["single"][flt]["results"][i][0] == out["single"][flt]["results"][i][0], (flt, i)
                assert np.allclose(ref["single"][flt]["results"][i][1], out["single"][flt]["results"][i][1], rtol=1e-4)


def test_gfx950_module_loads_on_cpu_and_refuses_to_search_without_a_gpu():
    """CPU: the drop-in module imports (the reference binding's 8 methods + rebuild + query_batch), ingests through the unchanged
    DBServer, and a query fails loudly when no gfx950 device is usable - there is no CPU fallback behind the binding either."""
    if not _have(GPU_DIR):
        pytest.skip("dropin/_build/epsilla.so not built (make -C dropin)")
    import torch
    code = r'''
import sys, ctypes
sys.path.insert(0, %r)
import epsilla
names = ["load_db", "unload_db", "use_db", "create_table", "insert", "query", "drop_table", "delete", "rebuild", "query_batch", "load_db_scaled"]
assert all(hasattr(epsilla, n) for n in names), [n for n in names if not hasattr(epsilla, n)]
assert epsilla.backend == "gfx950"
assert epsilla.load_db(db_name="db", db_path=sys.argv[1]) == 0
epsilla.use_db(db_name="db")
fields = [{"name": "ID", "dataType": "INT", "primaryKey": True}, {"name": "V", "dataType": "VECTOR_FLOAT", "dimensions": 4, "metricType": "EUCLIDEAN"}]
ctypes.pythonapi.Py_IncRef(ctypes.py_object(fields))
epsilla.create_table(table_name="T", table_fields=fields)
assert epsilla.insert(table_name="T", records=[{"ID": 1, "V": [0.1, 0.2, 0.3, 0.4]}]) == 0
try:
    r = epsilla.query(table_name="T", query_field="V", query_vector=[0.1, 0.2, 0.3, 0.4], response_fields=["ID"], limit=1, filter="", with_distance=True)
    print("ANSWERED", r)
except BaseException as e:
    print("REFUSED", e)
''' % GPU_DIR
    db = os.path.join(tempfile.mkdtemp(), "db")
    r = subprocess.run([sys.executable, "-c", code, db], capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    if torch.cuda.is_available():
        assert "ANSWERED" in out, out[-2000:]
    else:
        assert "ANSWERED" not in out and ("REFUSED" in out or r.returncode != 0), out[-2000:]
        assert "gfx950" in out, out[-2000:]


@pytest.mark.gpu
def test_gfx950_module_insert_array_leaves_the_table_the_json_insert_leaves():
    """insert_array() - records as column buffers - against the reference's own ingest (insert(): JSON ->
    TableMVP::Insert -> TableSegmentMVP::Insert, engine/db/table_segment_mvp.cpp:455-808) on the same records in the same module: 100 000
    records, three vector fields (EUCLIDEAN / DOT_PRODUCT / COSINE: rows normalised at insert exactly as :574-587, the zero vector left
    alone), INT / FLOAT / DOUBLE / BOOL / STRING attributes, duplicate primary keys inside a batch, across batches and at the very end,
    different batch boundaries on the two sides, deletes afterwards.  Every query() / query_batch() answer - ids, distances, attributes,
    the stored vectors themselves - is identical; capacity and missing-field errors read like the reference's; upsert replaces a row."""
    if not _have(GPU_DIR):
        pytest.skip("dropin/_build/epsilla.so not built (make -C dropin)")
    rows = 100_000
    out = _run(GPU_DIR, "ingest", rows, 48, 40)
    assert out["json_codes_ok"] and out["delete_codes"] == [0, 0]
    ins = sum(r[1]["inserted"] for r in out["array_results"])
    skp = sum(r[1]["skipped"] for r in out["array_results"])
    assert all(r[0] == 0 for r in out["array_results"]) and (ins, skp) == (rows - 2, 2), out["array_results"]
    J, A = out["answers"]["J"], out["answers"]["A"]
    assert set(J) == set(A) and len(J) == 18
    for key in J:
        assert J[key] == A[key], key                     # (JSON round trip of the SAME floats on both sides: equality, not closeness)
    some = J["query|VCOS|"][0][1]
    assert len(some) == 10 and abs(sum(v * v for v in some[0]["VCOS"]) - 1.0) < 1e-5
    # ... and after rebuild() (both tables' graphs built on their device mirrors): the graph path answers them alike too
    assert out["rebuild_code"] == 0
    RJ, RA = out["answers_after_rebuild"]["J"], out["answers_after_rebuild"]["A"]
    for field in ("VL2", "VIP", "VCOS"):
        assert RJ[field][0] == 0 and RJ[field] == RA[field], field
    assert out["capacity_error"] and "can hold up to 150000 records" in out["capacity_error"]
    assert out["missing_field_error"] and "missing field: Tag" in out["missing_field_error"]
    assert out["upsert"] == [0, {"inserted": 2, "skipped": 0}]
    assert [r["Name"] for r in out["after_upsert"]] == ["upserted", "new"] or out["after_upsert"][0]["Name"] == "upserted"
    assert out["array_insert_s"] < 0.2 * out["json_insert_s"], (out["array_insert_s"], out["json_insert_s"])


def test_gfx950_module_insert_array_argument_checks_on_cpu():
    """CPU (ingest never touches the device): what insert_array accepts and refuses - every schema field needs a column of a matching shape, numeric columns
    are C-contiguous buffers, STRING columns lists of str, all of one length; n = 0 is a no-op; float64 vectors and VECTOR_DOUBLE fields are stored as the
    JSON path stores them (rounded to float32); a primary key repeated inside the call is skipped as `insert` skips it."""
    if not _have(GPU_DIR):
        pytest.skip("dropin/_build/epsilla.so not built (make -C dropin)")
    code = r'''
import sys, ctypes
import numpy as np
sys.path.insert(0, %r)
import epsilla
assert epsilla.load_db(db_name="db", db_path=sys.argv[1]) == 0
epsilla.use_db(db_name="db")
fields = [{"name": "ID", "dataType": "INT", "primaryKey": True}, {"name": "Name", "dataType": "STRING"}, {"name": "W", "dataType": "DOUBLE"},
          {"name": "V", "dataType": "VECTOR_FLOAT", "dimensions": 4, "metricType": "COSINE"},
          {"name": "D", "dataType": "VECTOR_DOUBLE", "dimensions": 3, "metricType": "EUCLIDEAN"}]
ctypes.pythonapi.Py_IncRef(ctypes.py_object(fields))
epsilla.create_table(table_name="T", table_fields=fields)
n = 5
good = {"ID": np.array([1, 2, 3, 2, 5], np.int64), "Name": ["a", "b", "c", "dup", "e"], "W": np.arange(n, dtype=np.float32),
        "V": np.arange(4 * n, dtype=np.float64).reshape(n, 4) + 1.0, "D": np.ones((n, 3), np.float32)}
def refused(cols, **kw):
    try:
        epsilla.insert_array(table_name="T", columns=cols, **kw)
    except Exception as e:
        return str(e)
    return None
bad = []
bad.append(refused({k: v for k, v in good.items() if k != "W"}))                                  # a schema field without a column
bad.append(refused({**good, "V": good["V"][:, :3]}))                                               # wrong width
bad.append(refused({**good, "V": np.asfortranarray(good["V"])}))                                   # not C-contiguous
bad.append(refused({**good, "Name": ["a", "b", "c"]}))                                             # lengths differ
bad.append(refused({**good, "Name": np.arange(n)}))                                                # a STRING field wants a list of str
bad.append(refused({**good, "ID": np.arange(n, dtype=np.float64)}))                                # an integer field wants an integer column
bad.append(refused({**good, "ID": [1, 2, 3, 4, 5]}))                                               # numeric columns are buffers, not lists
bad.append(refused("not a dict"))
assert all(bad), bad
assert epsilla.insert_array(table_name="T", columns={"ID": np.zeros(0, np.int64), "Name": [], "W": np.zeros(0), "V": np.zeros((0, 4), np.float32), "D": np.zeros((0, 3))}) == (0, {"inserted": 0, "skipped": 0})
assert epsilla.insert_array(table_name="T", columns=good) == (0, {"inserted": 4, "skipped": 1})
assert epsilla.insert_array(table_name="T", columns=good) == (0, {"inserted": 0, "skipped": 5})
print("CHECKS OK", [b[:60] for b in bad])
''' % GPU_DIR
    db = os.path.join(tempfile.mkdtemp(), "db")
    r = subprocess.run([sys.executable, "-c", code, db], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "CHECKS OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
