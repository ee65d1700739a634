// The `epsilla` CPython module of the drop-in build (SURVEY 8b "Python-module ABI", 8f rank 1).
//
// The reference's binding (engine/bindings/python/interface.cpp + interface.h) is compiled UNMODIFIED: it is included
// below by path, with its module-init symbol renamed so that this file can extend the module it creates.  Its eight
// methods (load_db, unload_db, use_db, create_table, insert, query, drop_table, delete; interface.h:22-32) therefore keep
// their exact argument formats, return conventions and quirks (interface.cpp:274-287, :396).  What runs underneath is
// the reference's own DBServer / TableMVP over THIS repository's VecSearchExecutor / ANNGraphSegment (include/db/**,
// dropin/*.cpp -> libepsilla_gfx950.so), so every query() is answered on the MI355X.
//
// Additive entry points (the reference module has no way to trigger an index build and takes one vector per call):
//   rebuild() -> int                                   DBServer::Rebuild (db_server.hpp:112) as leader; the graph is built
//                                                      on the device by ANNGraphSegment::BuildFromVectorTable
//   load_db_scaled(db_name, db_path, vector_scale, wal_enabled=True) -> int
//                                                      load_db with the table capacity the REST API calls vectorScale
//   query_batch(table_name, query_field, query_vectors, response_fields, limit, filter, with_distance)
//       [, as_arrays=False]                            (r5) as_arrays=True: (int, {field: ndarray[N][limit], "@distance": float32[N][limit], "@count": int32[N]})
//       -> (int, list[list[dict]])                     query() for N vectors (a 2-D float32 / float64 buffer such as a NumPy
//                                                      array, or a list of lists): ONE eps_index_search with nq = N through
//                                                      VecSearchExecutor::SearchBatch, one projection pass; element q of the
//                                                      result equals query(..., query_vectors[q], ...)[1]
//   insert_array(table_name, columns, upsert=False, sync=False) -> (int, {"inserted": n, "skipped": m})
//                                                      (r6) insert() for n records given as COLUMNS - {field: 2-D float32 / float64 buffer}
//                                                      for dense vector fields, {field: 1-D integer / float buffer} for primitive
//                                                      fields, {field: list of str} for STRING fields: what TableSegmentMVP::Insert
//                                                      would have stored for the same records (COSINE normalisation, duplicate
//                                                      primary keys, capacity check), without a JSON document or a text WAL record
//                                                      (epsdrop::InsertArray, include/epsdrop/insert_array.hpp)
#define PyInit_epsilla PyInit_epsilla_reference_binding
#include "epsdrop/insert_array.hpp"
#include "epsdrop/search_batch.hpp"
#include "bindings/python/interface.cpp"  // the reference's binding, from where it lies under $(REF)
#undef PyInit_epsilla

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <thread>

static PyObject* eps_rebuild(PyObject* self, PyObject* args, PyObject* kwargs) {
  (void)self;
  (void)args;
  (void)kwargs;
  int code = 0;
  std::string err;
  Py_BEGIN_ALLOW_THREADS
  try {
    db->SetLeader(true);   // Rebuild is a no-op on followers (db_server.cpp); the reference's tests do the same
    code = db->Rebuild().code();
  } catch (const std::exception& e) {
    err = e.what();
  }
  Py_END_ALLOW_THREADS
  if (!err.empty()) {
    PyErr_SetString(PyExc_Exception, err.c_str());
    return NULL;
  }
  return PyLong_FromLong(code);
}

// load_db with the table capacity as an argument.  The REST entry takes it from the request ("vectorScale",
// server/web_server/web_controller.hpp:120-140); the binding hard-codes 150000 (interface.cpp:55-61), and the segment loader
// refuses files with more records than that (table_segment_mvp.cpp:161-166).
static PyObject* eps_load_db_scaled(PyObject* self, PyObject* args, PyObject* kwargs) {
  (void)self;
  static const char* keywords[] = {"db_name", "db_path", "vector_scale", "wal_enabled", NULL};
  const char *namePtr, *pathPtr;
  long long scale = 150000;
  int wal = 1;
  if (!PyArg_ParseTupleAndKeywords(args, kwargs, "ssL|p", (char**)keywords, &namePtr, &pathPtr, &scale, &wal)) return NULL;
  const std::string name = namePtr, path = pathPtr;
  if (name.empty() || path.empty() || scale <= 0) {
    PyErr_SetString(PyExc_Exception, "load_db_scaled: empty db name / path, or vector_scale <= 0");
    return NULL;
  }
  int code = 0;
  std::string err;
  Py_BEGIN_ALLOW_THREADS
  try {
    std::unordered_map<std::string, std::string> headers;
    code = db->LoadDB(name, path, (int64_t)scale, wal != 0, headers).code();
  } catch (const std::exception& e) {
    err = e.what();
  }
  Py_END_ALLOW_THREADS
  if (!err.empty()) {
    PyErr_SetString(PyExc_Exception, err.c_str());
    return NULL;
  }
  return PyLong_FromLong(code);
}

// ---- query_batch: ONE device batch for all the vectors -------------------------------------------------------------------
// What DBServer::Search (db/db_server.cpp:458-510) and TableMVP::Search (db/table_mvp.cpp:300-399) do per vector is done once
// for the batch - table / field lookup and validation, filter parsing, query normalisation for COSINE, one executor from the
// field's pool - then VecSearchExecutor::SearchBatch (one eps_index_search with nq = N) and one projection pass.
namespace {

struct FieldPlan {   // how to turn one response field of one row into a Python object without a JSON round trip
  std::string name;
  vectordb::engine::meta::FieldType type;
  size_t offset = 0;   // byte offset inside the attribute row / index of the variable-length column / of the vector table
  int64_t dim = 0;
  PyObject* key = nullptr;
};

// the projection of TableMVP::Project (db/table_mvp.cpp:462-583) for primitive, string and dense-vector fields, built directly as
// Python objects; fields it does not cover (JSON, GEO_POINT, sparse vectors) make the caller use Project + json.loads instead
bool PlanFields(vectordb::engine::TableMVP& table, std::vector<std::string>& fields, std::vector<FieldPlan>& plan) {
  using vectordb::engine::meta::FieldType;
  if (fields.empty())
    for (auto& f : table.table_schema_.fields_)
      if (!f.is_index_field_) fields.push_back(f.name_);
  auto& seg = *table.table_segment_;
  for (auto& name : fields) {
    FieldPlan p;
    p.name = name;
    p.type = table.field_name_field_type_map_[name];
    p.offset = seg.field_name_mem_offset_map_[name];
    switch (p.type) {
      case FieldType::INT1: case FieldType::INT2: case FieldType::INT4: case FieldType::INT8: case FieldType::FLOAT:
      case FieldType::DOUBLE: case FieldType::BOOL: case FieldType::STRING:
        break;
      case FieldType::VECTOR_FLOAT: case FieldType::VECTOR_DOUBLE:
        p.dim = seg.vector_dims_[p.offset];
        break;
      default:
        return false;
    }
    plan.push_back(p);
  }
  return true;
}

PyObject* RowToDict(vectordb::engine::TableSegmentMVP& seg, const std::vector<FieldPlan>& plan, int64_t id, bool with_distance, double distance,
                    PyObject* dist_key) {
  using vectordb::engine::meta::FieldType;
  PyObject* d = PyDict_New();
  if (!d) return NULL;
  for (auto& p : plan) {
    const char* at = seg.attribute_table_ + p.offset + id * seg.primitive_offset_;
    PyObject* v = NULL;
    switch (p.type) {
      case FieldType::INT1: { int8_t x; std::memcpy(&x, at, 1); v = PyLong_FromLongLong(x); break; }
      case FieldType::INT2: { int16_t x; std::memcpy(&x, at, 2); v = PyLong_FromLongLong(x); break; }
      case FieldType::INT4: { int32_t x; std::memcpy(&x, at, 4); v = PyLong_FromLongLong(x); break; }
      case FieldType::INT8: { int64_t x; std::memcpy(&x, at, 8); v = PyLong_FromLongLong(x); break; }
      case FieldType::FLOAT: { float x; std::memcpy(&x, at, 4); v = PyFloat_FromDouble((double)x); break; }
      case FieldType::DOUBLE: { double x; std::memcpy(&x, at, 8); v = PyFloat_FromDouble(x); break; }
      case FieldType::BOOL: { bool x; std::memcpy(&x, at, 1); v = PyBool_FromLong(x ? 1 : 0); break; }
      case FieldType::STRING: {
        const std::string& str = std::get<std::string>(seg.var_len_attr_table_[p.offset][id]);
        v = PyUnicode_FromStringAndSize(str.data(), (Py_ssize_t)str.size());
        break;
      }
      default: {   // dense vector
        v = PyList_New((Py_ssize_t)p.dim);
        if (v)
          for (int64_t k = 0; k < p.dim; ++k) PyList_SET_ITEM(v, (Py_ssize_t)k, PyFloat_FromDouble((double)seg.vector_tables_[p.offset][id * p.dim + k]));
      }
    }
    if (!v || PyDict_SetItem(d, p.key, v) < 0) {
      Py_XDECREF(v);
      Py_DECREF(d);
      return NULL;
    }
    Py_DECREF(v);
  }
  if (with_distance) {
    PyObject* v = PyFloat_FromDouble(distance);
    if (!v || PyDict_SetItem(d, dist_key, v) < 0) {
      Py_XDECREF(v);
      Py_DECREF(d);
      return NULL;
    }
    Py_DECREF(v);
  }
  return d;
}

}  // namespace

static PyObject* eps_query_batch(PyObject* self, PyObject* args, PyObject* kwargs) {
  (void)self;
  static const char* keywords[] = {"table_name", "query_field", "query_vectors", "response_fields", "limit", "filter", "with_distance", "as_arrays", NULL};
  const char *tableNamePtr, *queryFieldPtr, *queryFilterPtr;
  int limit, withDistance, asArrays = 0;
  PyObject *queryVectors, *responseFields;
  if (!PyArg_ParseTupleAndKeywords(args, kwargs, "ssOOisp|p", (char**)keywords, &tableNamePtr, &queryFieldPtr, &queryVectors, &responseFields,
                                   &limit, &queryFilterPtr, &withDistance, &asArrays))
    return NULL;
  if (!PyList_Check(responseFields)) {
    PyErr_SetString(PyExc_Exception, "response_fields must be a list");
    return NULL;
  }
  // ---- the query matrix: any C-contiguous 2-D buffer of float32 / float64 (NumPy array, torch CPU tensor, memoryview), or a
  // list of lists of float
  std::vector<float> qbuf;
  const float* qptr = nullptr;
  Py_ssize_t nq = 0, dim = 0;
  Py_buffer view;
  bool have_view = false;
  if (PyObject_CheckBuffer(queryVectors) && PyObject_GetBuffer(queryVectors, &view, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) == 0) {
    have_view = true;
    const bool f32 = view.format && (std::strcmp(view.format, "f") == 0 || std::strcmp(view.format, "<f") == 0 || std::strcmp(view.format, "=f") == 0);
    const bool f64 = view.format && (std::strcmp(view.format, "d") == 0 || std::strcmp(view.format, "<d") == 0 || std::strcmp(view.format, "=d") == 0);
    if (view.ndim != 2 || !(f32 || f64)) {
      PyBuffer_Release(&view);
      PyErr_SetString(PyExc_Exception, "query_vectors: a 2-D C-contiguous float32 / float64 buffer, or a list of lists of float");
      return NULL;
    }
    nq = view.shape[0];
    dim = view.shape[1];
    if (f32) {
      qptr = static_cast<const float*>(view.buf);
    } else {
      qbuf.resize((size_t)nq * dim);
      const double* src = static_cast<const double*>(view.buf);
      for (size_t i = 0; i < qbuf.size(); ++i) qbuf[i] = (float)src[i];
      qptr = qbuf.data();
    }
  } else {
    PyErr_Clear();
    if (!PyList_Check(queryVectors)) {
      PyErr_SetString(PyExc_Exception, "query_vectors: a 2-D C-contiguous float32 / float64 buffer, or a list of lists of float");
      return NULL;
    }
    nq = PyList_Size(queryVectors);
    for (Py_ssize_t q = 0; q < nq; ++q) {
      PyObject* v = PyList_GetItem(queryVectors, q);
      if (!PyList_Check(v) || (q > 0 && PyList_Size(v) != dim)) {
        PyErr_SetString(PyExc_Exception, "query_vectors must be a list of lists of float of one length");
        return NULL;
      }
      if (q == 0) {
        dim = PyList_Size(v);
        qbuf.resize((size_t)nq * dim);
      }
      for (Py_ssize_t i = 0; i < dim; ++i) qbuf[(size_t)q * dim + i] = (float)PyFloat_AsDouble(PyList_GetItem(v, i));
    }
    if (PyErr_Occurred()) return NULL;
    qptr = qbuf.data();
  }
  struct ViewGuard {
    Py_buffer* v;
    ~ViewGuard() { if (v) PyBuffer_Release(v); }
  } guard{have_view ? &view : nullptr};

  std::vector<std::string> fields;
  for (Py_ssize_t i = 0; i < PyList_Size(responseFields); ++i) {
    PyObject* s = PyObject_Str(PyList_GetItem(responseFields, i));
    if (!s) return NULL;
    fields.push_back(PyUnicode_AsUTF8(s));
    Py_DECREF(s);
  }
  const std::string tableName = tableNamePtr, queryFilter = queryFilterPtr;
  std::string fieldName = queryFieldPtr;

  // ---- lookups, validation, COSINE normalisation, executor and ONE device batch: the C++-level entry any host can call
  // (epsdrop::SearchBatch, include/epsdrop/search_batch.hpp)
  std::string err;
  epsdrop::BatchHits hits;
  static const bool timing = getenv("EPS_DROPIN_TIMING") && atoi(getenv("EPS_DROPIN_TIMING")) != 0;   // (phase times of every call on stderr)
  const auto t_args = std::chrono::steady_clock::now();
  Py_BEGIN_ALLOW_THREADS
  try {
    for (auto& f : fields) {   // (TableMVP::Search checks the response fields before it searches, table_mvp.cpp:310-314)
      auto database = db->GetDB(db_name);
      auto t = database ? database->GetTable(tableName) : nullptr;
      if (t && t->field_name_field_type_map_.find(f) == t->field_name_field_type_map_.end()) throw std::runtime_error("Field name not found: " + f);
    }
    const vectordb::Status st = epsdrop::SearchBatch(*db, db_name, tableName, fieldName, qptr, (int64_t)nq, (int64_t)dim, (int64_t)limit, queryFilter, &hits);
    if (!st.ok()) throw std::runtime_error(st.message().empty() ? std::string("query_batch failed") : st.message());
  } catch (const std::exception& e) {
    err = e.what();
    if (err.empty()) err = "query_batch failed";
  }
  Py_END_ALLOW_THREADS
  if (!err.empty()) {
    PyErr_SetString(PyExc_Exception, err.c_str());
    return NULL;
  }
  const auto t_search = std::chrono::steady_clock::now();
  std::shared_ptr<vectordb::engine::TableMVP> table = hits.table;
  std::vector<int64_t>& ids = hits.ids;
  std::vector<float>& dist = hits.dist;
  std::vector<int32_t>& counts = hits.counts;
  const int32_t width = hits.width;

  // ---- as_arrays=True (r5): the answer as NumPy arrays instead of nq x limit dicts - one [nq][limit] array per requested numeric field,
  // "@distance" float32 [nq][limit] (+inf beyond a query's count), "@count" int32 [nq].  Creating and later freeing 10 240 dicts is 1.1 ms of
  // an 8.6 ms batch at 10M x 768 (profiles/r5_module_query_batch_10Mx768_phases.txt); the arrays cost microseconds.
  if (asArrays) {
    using vectordb::engine::meta::FieldType;
    std::vector<FieldPlan> plan;
    if (!PlanFields(*table, fields, plan)) {
      PyErr_SetString(PyExc_Exception, "query_batch(as_arrays=True): response_fields must be numeric or BOOL fields");
      return NULL;
    }
    PyObject* np = PyImport_ImportModule("numpy");
    if (!np) return NULL;
    PyObject* result = PyDict_New();
    auto& seg = *table->table_segment_;
    const Py_ssize_t total = (Py_ssize_t)nq * limit;
    auto put = [&](const char* key, const void* data, size_t bytes, const char* dtype, bool matrix) -> bool {
      PyObject* b = PyBytes_FromStringAndSize(static_cast<const char*>(data), (Py_ssize_t)bytes);
      PyObject* a = b ? PyObject_CallMethod(np, "frombuffer", "Os", b, dtype) : NULL;
      Py_XDECREF(b);
      if (a && matrix) {
        PyObject* r = PyObject_CallMethod(a, "reshape", "nn", (Py_ssize_t)nq, (Py_ssize_t)limit);
        Py_DECREF(a);
        a = r;
      }
      const bool ok2 = a && PyDict_SetItemString(result, key, a) == 0;
      Py_XDECREF(a);
      return ok2;
    };
    bool ok2 = result != NULL;
    std::vector<int32_t> cnts((size_t)nq);
    for (Py_ssize_t q = 0; q < nq; ++q) cnts[(size_t)q] = (int32_t)std::max<int64_t>(0, std::min<int64_t>(counts[(size_t)q], limit));
    for (auto& p : plan) {
      if (!ok2) break;
      if (p.type == FieldType::STRING || p.type == FieldType::VECTOR_FLOAT || p.type == FieldType::VECTOR_DOUBLE) {
        PyErr_SetString(PyExc_Exception, "query_batch(as_arrays=True): response_fields must be numeric or BOOL fields");
        ok2 = false;
        break;
      }
      const bool is_float = p.type == FieldType::FLOAT || p.type == FieldType::DOUBLE;
      std::vector<int64_t> iv(is_float ? 0 : (size_t)total, -1);
      std::vector<double> fv(is_float ? (size_t)total : 0, std::numeric_limits<double>::quiet_NaN());
      for (Py_ssize_t q = 0; q < nq; ++q)
        for (int32_t i = 0; i < cnts[(size_t)q]; ++i) {
          const char* at = seg.attribute_table_ + p.offset + ids[(size_t)q * width + i] * seg.primitive_offset_;
          const size_t o = (size_t)q * limit + i;
          switch (p.type) {
            case FieldType::INT1: { int8_t x; std::memcpy(&x, at, 1); iv[o] = x; break; }
            case FieldType::INT2: { int16_t x; std::memcpy(&x, at, 2); iv[o] = x; break; }
            case FieldType::INT4: { int32_t x; std::memcpy(&x, at, 4); iv[o] = x; break; }
            case FieldType::INT8: { int64_t x; std::memcpy(&x, at, 8); iv[o] = x; break; }
            case FieldType::BOOL: { bool x; std::memcpy(&x, at, 1); iv[o] = x ? 1 : 0; break; }
            case FieldType::FLOAT: { float x; std::memcpy(&x, at, 4); fv[o] = x; break; }
            default: { double x; std::memcpy(&x, at, 8); fv[o] = x; }
          }
        }
      ok2 = is_float ? put(p.name.c_str(), fv.data(), fv.size() * 8, "float64", true) : put(p.name.c_str(), iv.data(), iv.size() * 8, "int64", true);
    }
    if (ok2 && withDistance) {
      std::vector<float> dv((size_t)total, std::numeric_limits<float>::infinity());
      for (Py_ssize_t q = 0; q < nq; ++q)
        for (int32_t i = 0; i < cnts[(size_t)q]; ++i) dv[(size_t)q * limit + i] = dist[(size_t)q * width + i];
      ok2 = put("@distance", dv.data(), dv.size() * 4, "float32", true);
    }
    if (ok2) ok2 = put("@count", cnts.data(), cnts.size() * 4, "int32", false);
    Py_DECREF(np);
    if (!ok2) {
      Py_XDECREF(result);
      if (!PyErr_Occurred()) PyErr_SetString(PyExc_Exception, "query_batch(as_arrays=True): could not build the arrays");
      return NULL;
    }
    if (timing) {
      const auto t_end = std::chrono::steady_clock::now();
      fprintf(stderr, "[epsilla.query_batch] %lld vectors: search (lookups + H2D + device + D2H) %.3f ms, arrays %.3f ms\n", (long long)nq,
              1e3 * std::chrono::duration<double>(t_search - t_args).count(), 1e3 * std::chrono::duration<double>(t_end - t_search).count());
    }
    return Py_BuildValue("(iN)", 0, result);
  }

  // ---- projection, once for the whole batch
  PyObject* out = PyList_New(nq);
  if (!out) return NULL;
  std::vector<FieldPlan> plan;
  const bool direct = PlanFields(*table, fields, plan);
  PyObject* dist_key = PyUnicode_FromString("@distance");
  PyObject* loads = NULL;
  if (direct) {
    for (auto& p : plan) p.key = PyUnicode_FromString(p.name.c_str());
  } else {
    PyObject* json_module = PyImport_ImportModule("json");
    loads = json_module ? PyObject_GetAttrString(json_module, "loads") : NULL;
    Py_XDECREF(json_module);
  }
  bool ok = dist_key && (direct || loads);
  for (Py_ssize_t q = 0; ok && q < nq; ++q) {
    const int64_t cnt = std::max<int64_t>(0, std::min<int64_t>(q < (Py_ssize_t)counts.size() ? counts[(size_t)q] : 0, limit));
    PyObject* rows = NULL;
    if (direct) {
      rows = PyList_New((Py_ssize_t)cnt);
      for (int64_t i = 0; rows && i < cnt; ++i) {
        PyObject* d = RowToDict(*table->table_segment_, plan, ids[(size_t)q * width + i], withDistance != 0, (double)dist[(size_t)q * width + i], dist_key);
        if (!d) {
          Py_CLEAR(rows);
          break;
        }
        PyList_SET_ITEM(rows, (Py_ssize_t)i, d);
      }
    } else {   // the reference's own projection + its JSON round trip (interface.cpp:333-360)
      std::vector<int64_t> qi(ids.begin() + (size_t)q * width, ids.begin() + (size_t)q * width + cnt);
      std::vector<double> qd(dist.begin() + (size_t)q * width, dist.begin() + (size_t)q * width + cnt);
      vectordb::Json result;
      std::vector<std::string> f = fields;
      auto st = table->Project(f, cnt, qi, result, withDistance != 0, qd);
      if (!st.ok()) PyErr_SetString(PyExc_Exception, st.message().c_str());
      else rows = PyObject_CallFunction(loads, "s", result.DumpToString().c_str());
    }
    if (!rows) ok = false; else PyList_SET_ITEM(out, q, rows);
  }
  for (auto& p : plan) Py_XDECREF(p.key);
  Py_XDECREF(dist_key);
  Py_XDECREF(loads);
  if (!ok) {
    Py_DECREF(out);
    if (!PyErr_Occurred()) PyErr_SetString(PyExc_Exception, "query_batch: projection failed");
    return NULL;
  }
  if (timing) {
    const auto t_end = std::chrono::steady_clock::now();
    fprintf(stderr, "[epsilla.query_batch] %lld vectors: search (lookups + H2D + device + D2H) %.3f ms, projection to Python objects %.3f ms\n", (long long)nq,
            1e3 * std::chrono::duration<double>(t_search - t_args).count(), 1e3 * std::chrono::duration<double>(t_end - t_search).count());
  }
  return Py_BuildValue("(iN)", 0, out);
}

// ---- insert_array: n records as column buffers (epsdrop::InsertArray) -----------------------------------------------------
static PyObject* eps_insert_array(PyObject* self, PyObject* args, PyObject* kwargs) {
  (void)self;
  static const char* keywords[] = {"table_name", "columns", "upsert", "sync", NULL};
  const char* tableNamePtr;
  PyObject* columns;
  int upsert = 0, sync = 0;
  if (!PyArg_ParseTupleAndKeywords(args, kwargs, "sO|pp", (char**)keywords, &tableNamePtr, &columns, &upsert, &sync)) return NULL;
  if (!PyDict_Check(columns)) {
    PyErr_SetString(PyExc_Exception, "insert_array: columns must be a dict {field name: buffer | list of str}");
    return NULL;
  }
  struct Held {
    std::vector<Py_buffer> views;
    ~Held() { for (auto& v : views) PyBuffer_Release(&v); }
  } held;
  held.views.reserve((size_t)PyDict_Size(columns));
  std::vector<epsdrop::ColumnView> cols;
  std::vector<std::unique_ptr<std::vector<std::string>>> strs;
  int64_t n = -1;
  PyObject *key, *value;
  Py_ssize_t pos = 0;
  auto bad = [&](const std::string& what) -> PyObject* {
    PyErr_SetString(PyExc_Exception, ("insert_array: " + what).c_str());
    return NULL;
  };
  while (PyDict_Next(columns, &pos, &key, &value)) {
    if (!PyUnicode_Check(key)) return bad("column names must be str");
    epsdrop::ColumnView c;
    c.name = PyUnicode_AsUTF8(key);
    int64_t rows = 0;
    if (PyList_Check(value)) {   // a STRING column
      rows = (int64_t)PyList_Size(value);
      strs.emplace_back(new std::vector<std::string>());
      strs.back()->reserve((size_t)rows);
      for (Py_ssize_t i = 0; i < rows; ++i) {
        PyObject* s = PyList_GET_ITEM(value, i);
        if (!PyUnicode_Check(s)) return bad("column " + c.name + ": a list column holds str (numeric columns are buffers, e.g. NumPy arrays)");
        Py_ssize_t len = 0;
        const char* u = PyUnicode_AsUTF8AndSize(s, &len);
        if (!u) return NULL;
        strs.back()->emplace_back(u, (size_t)len);
      }
      c.kind = epsdrop::ColumnView::STR;
      c.strings = strs.back().get();
    } else {
      Py_buffer view;
      if (!PyObject_CheckBuffer(value) || PyObject_GetBuffer(value, &view, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) {
        PyErr_Clear();
        return bad("column " + c.name + ": a C-contiguous 1-D / 2-D buffer (NumPy array) or a list of str");
      }
      held.views.push_back(view);
      const char* f = view.format ? view.format : "B";
      if (*f == '<' || *f == '=' || *f == '@') ++f;
      const std::string fmt = f;
      if (fmt == "b") c.kind = epsdrop::ColumnView::I8;
      else if (fmt == "h") c.kind = epsdrop::ColumnView::I16;
      else if (fmt == "i" || (fmt == "l" && view.itemsize == 4)) c.kind = epsdrop::ColumnView::I32;
      else if (fmt == "q" || (fmt == "l" && view.itemsize == 8)) c.kind = epsdrop::ColumnView::I64;
      else if (fmt == "B" || fmt == "?") c.kind = epsdrop::ColumnView::U8;
      else if (fmt == "f") c.kind = epsdrop::ColumnView::F32;
      else if (fmt == "d") c.kind = epsdrop::ColumnView::F64;
      else return bad("column " + c.name + ": element type '" + fmt + "' (int8/16/32/64, uint8, bool, float32, float64)");
      if (view.ndim != 1 && view.ndim != 2) return bad("column " + c.name + ": 1-D (primitive field) or 2-D (dense vector field)");
      rows = (int64_t)view.shape[0];
      c.width = view.ndim == 2 ? (int64_t)view.shape[1] : 1;
      c.data = view.buf;
    }
    if (n >= 0 && rows != n) return bad("columns of different lengths");
    n = rows;
    cols.push_back(c);
  }
  if (n < 0) n = 0;
  const std::string tableName = tableNamePtr;
  epsdrop::InsertArrayResult r;
  int code = 0;
  std::string err;
  Py_BEGIN_ALLOW_THREADS
  try {
    const vectordb::Status st = epsdrop::InsertArray(*db, db_name, tableName, cols, n, upsert != 0, sync != 0, &r);
    code = st.code();
    if (!st.ok()) err = st.message().empty() ? std::string("insert_array failed") : st.message();
  } catch (const std::exception& e) {
    err = e.what();
    if (err.empty()) err = "insert_array failed";
  }
  Py_END_ALLOW_THREADS
  if (!err.empty()) {
    PyErr_SetString(PyExc_Exception, err.c_str());
    return NULL;
  }
  return Py_BuildValue("(i{s:L,s:L})", code, "inserted", (long long)r.inserted, "skipped", (long long)r.skipped);
}

static PyMethodDef EpsillaGfx950Methods[] = {
    {"rebuild", (PyCFunction)(void (*)(void))eps_rebuild, METH_VARARGS | METH_KEYWORDS, "build the ANN graphs now (additive: DBServer::Rebuild as leader)"},
    {"query_batch", (PyCFunction)(void (*)(void))eps_query_batch, METH_VARARGS | METH_KEYWORDS, "query() for N vectors in one device batch (additive)"},
    {"insert_array", (PyCFunction)(void (*)(void))eps_insert_array, METH_VARARGS | METH_KEYWORDS, "insert() for n records given as column buffers (additive; no JSON, no text WAL record)"},
    {"load_db_scaled", (PyCFunction)(void (*)(void))eps_load_db_scaled, METH_VARARGS | METH_KEYWORDS, "load_db with the table capacity (the REST API's vectorScale) as an argument (additive)"},
    {NULL, NULL, 0, NULL}};

PyMODINIT_FUNC PyInit_epsilla(void) {
  PyObject* m = PyInit_epsilla_reference_binding();
  if (!m) return NULL;
  if (PyModule_AddFunctions(m, EpsillaGfx950Methods) < 0) {
    Py_DECREF(m);
    return NULL;
  }
  PyModule_AddStringConstant(m, "backend", "gfx950");
  return m;
}
