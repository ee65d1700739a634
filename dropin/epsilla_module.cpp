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
//   query_batch(table_name, query_field, query_vectors, response_fields, limit, filter, with_distance)
//       -> (int, list[list[dict]])                     the same as query() for a list of vectors: the calls are issued
//                                                      concurrently (GIL released) through the unchanged
//                                                      DBServer::Search, and the executor's micro-batcher
//                                                      (dropin/vec_search_executor.cpp) coalesces them into device batches
#define PyInit_epsilla PyInit_epsilla_reference_binding
#include "bindings/python/interface.cpp"  // the reference's binding, from where it lies under $(REF)
#undef PyInit_epsilla

#include <atomic>
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

static PyObject* eps_query_batch(PyObject* self, PyObject* args, PyObject* kwargs) {
  (void)self;
  static const char* keywords[] = {"table_name", "query_field", "query_vectors", "response_fields", "limit", "filter", "with_distance", "threads", NULL};
  const char *tableNamePtr, *queryFieldPtr, *queryFilterPtr;
  int limit, withDistance, threads = 64;
  PyObject *queryVectors, *responseFields;
  if (!PyArg_ParseTupleAndKeywords(args, kwargs, "ssOOisp|i", (char**)keywords, &tableNamePtr, &queryFieldPtr, &queryVectors, &responseFields,
                                   &limit, &queryFilterPtr, &withDistance, &threads))
    return NULL;
  if (!PyList_Check(queryVectors) || !PyList_Check(responseFields)) {
    PyErr_SetString(PyExc_Exception, "query_vectors and response_fields must be lists");
    return NULL;
  }
  const Py_ssize_t nq = PyList_Size(queryVectors);
  std::vector<std::vector<float>> vecs((size_t)nq);
  for (Py_ssize_t q = 0; q < nq; ++q) {
    PyObject* v = PyList_GetItem(queryVectors, q);
    if (!PyList_Check(v)) {
      PyErr_SetString(PyExc_Exception, "query_vectors must be a list of lists of float");
      return NULL;
    }
    const Py_ssize_t d = PyList_Size(v);
    vecs[q].resize((size_t)d);
    for (Py_ssize_t i = 0; i < d; ++i) vecs[q][i] = (float)PyFloat_AsDouble(PyList_GetItem(v, i));
  }
  if (PyErr_Occurred()) return NULL;
  std::vector<std::string> fields;
  for (Py_ssize_t i = 0; i < PyList_Size(responseFields); ++i) {
    PyObject* s = PyObject_Str(PyList_GetItem(responseFields, i));
    fields.push_back(PyUnicode_AsUTF8(s));
    Py_XDECREF(s);
  }
  const std::string tableName = tableNamePtr, queryField = queryFieldPtr, queryFilter = queryFilterPtr;
  std::vector<std::string> results((size_t)nq);
  std::vector<int> codes((size_t)nq, 0);
  std::vector<std::string> errors((size_t)nq);
  Py_BEGIN_ALLOW_THREADS
  std::atomic<Py_ssize_t> next{0};
  auto worker = [&]() {
    for (;;) {
      const Py_ssize_t q = next.fetch_add(1);
      if (q >= nq) break;
      try {
        auto result = vectordb::Json();
        auto facetsConfig = vectordb::Json();
        facetsConfig.LoadFromString("[]");
        auto facets = vectordb::Json();
        std::vector<std::string> f = fields;
        std::string fieldName = queryField;
        auto status = db->Search(db_name, tableName, fieldName, f, (int64_t)vecs[q].size(), vecs[q].data(), limit, result, queryFilter,
                                 withDistance != 0, facetsConfig, facets);
        codes[q] = status.code();
        if (status.ok()) results[q] = result.DumpToString(); else errors[q] = status.message();
      } catch (const std::exception& e) {
        codes[q] = -1;
        errors[q] = e.what();
      }
    }
  };
  const int nt = (int)std::max<Py_ssize_t>(1, std::min<Py_ssize_t>(nq, threads));
  std::vector<std::thread> pool;
  for (int t = 0; t < nt; ++t) pool.emplace_back(worker);
  for (auto& t : pool) t.join();
  Py_END_ALLOW_THREADS
  for (Py_ssize_t q = 0; q < nq; ++q)
    if (codes[q] != 0) {
      PyErr_SetString(PyExc_Exception, errors[q].c_str());
      return NULL;
    }
  PyObject* json_module = PyImport_ImportModule("json");
  if (!json_module) return NULL;
  PyObject* loads = PyObject_GetAttrString(json_module, "loads");
  Py_DECREF(json_module);
  if (!loads) return NULL;
  PyObject* out = PyList_New(nq);
  for (Py_ssize_t q = 0; q < nq; ++q) {
    PyObject* r = PyObject_CallFunction(loads, "s", results[q].c_str());
    if (!r) {
      Py_DECREF(loads);
      Py_DECREF(out);
      return NULL;
    }
    PyList_SetItem(out, q, r);
  }
  Py_DECREF(loads);
  return Py_BuildValue("(iN)", 0, out);
}

static PyMethodDef EpsillaGfx950Methods[] = {
    {"rebuild", (PyCFunction)(void (*)(void))eps_rebuild, METH_VARARGS | METH_KEYWORDS, "build the ANN graphs now (additive: DBServer::Rebuild as leader)"},
    {"query_batch", (PyCFunction)(void (*)(void))eps_query_batch, METH_VARARGS | METH_KEYWORDS, "query() for a list of vectors (additive)"},
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
