// C entry points over the DBMS layers (DBServer at the JSON level) for callers without a C++ toolchain: the Python tests and drivers
// (ctypes), scripts/bench_dropin_mt.py.  Every function forwards to vectordb::engine::DBServer - the reference's class, compiled
// unmodified - and nothing else.  The same file is compiled twice:
//   * into dropin/_build/libepsilla_dropin.so (-DEPS_DROPIN): the reference's DBServer on top of THIS repository's
//     VecSearchExecutor / ANNGraphSegment / GetDistFunc and libepsilla_gfx950.so;
//   * into oracle/_ref/libepsilla_ref.so: the same DBServer on top of the reference's own executor (the checker).
// so that one test drives both through identical calls.
#include <omp.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "config/config.hpp"
#include "db/db_server.hpp"
#ifdef EPS_DROPIN
#include "epsdrop/insert_array.hpp"
#include "epsdrop/search_batch.hpp"
#endif

extern "C" {


// ---------------------------------------------------------------- DBServer (JSON level)
void ref_config(int intra_query_threads, int search_queue_size, int rebuild_threads, int prefilter, int executors) {
  auto& c = vectordb::globalConfig;
  if (intra_query_threads > 0) c.setIntraQueryThreads(intra_query_threads);
  if (search_queue_size > 0) c.setSearchQueueSize(search_queue_size);
  if (rebuild_threads > 0) c.setRebuildThreads(rebuild_threads);
  if (prefilter >= 0) c.PreFilter.store(prefilter != 0);
  if (executors > 0) c.setNumExecutorPerField(executors);
}
// DBServer::is_leader_ is not initialised by the constructor; the reference's own tests call
// SetLeader(true) before LoadDB whenever they Rebuild (test/engine/db/db_server.cpp:810,946,1088).
void* ref_db_new() {
  auto* s = new vectordb::engine::DBServer();
  s->SetLeader(true);
  return s;
}
void ref_db_set_leader(void* h, int leader) { static_cast<vectordb::engine::DBServer*>(h)->SetLeader(leader != 0); }
void ref_db_free(void* h) { delete static_cast<vectordb::engine::DBServer*>(h); }
int ref_db_load(void* h, const char* name, const char* path, int64_t scale, int wal) {
  std::unordered_map<std::string, std::string> headers;
  return static_cast<vectordb::engine::DBServer*>(h)->LoadDB(name, path, scale, wal != 0, headers).code();
}
int ref_db_create_table(void* h, const char* db, const char* schema_json) {
  size_t id = 0;
  return static_cast<vectordb::engine::DBServer*>(h)->CreateTable(db, std::string(schema_json), id).code();
}
int ref_db_insert(void* h, const char* db, const char* table, const char* records_json) {
  vectordb::Json j;
  if (!j.LoadFromString(records_json)) return -1;
  std::unordered_map<std::string, std::string> headers;
  return static_cast<vectordb::engine::DBServer*>(h)->Insert(db, table, j, headers).code();
}
int ref_db_delete(void* h, const char* db, const char* table, const char* pk_json, const char* filter) {
  vectordb::Json j;
  if (!j.LoadFromString(pk_json)) return -1;
  return static_cast<vectordb::engine::DBServer*>(h)->Delete(db, table, j, filter).code();
}
int ref_db_rebuild(void* h) {
  try {
    return static_cast<vectordb::engine::DBServer*>(h)->Rebuild().code();
  } catch (const std::exception& e) {
    fprintf(stderr, "Rebuild failed: %s\n", e.what());
    return vectordb::INFRA_UNEXPECTED_ERROR;
  }
}
int ref_db_swap_executors(void* h) { return static_cast<vectordb::engine::DBServer*>(h)->SwapExecutors().code(); }
// fields_csv: comma separated response fields.  Result JSON is written into out (NUL terminated,
// truncated to cap).  Returns the Status code.
int ref_db_search(void* h, const char* db, const char* table, const char* field, const char* fields_csv, float* q,
                  int64_t d, int64_t limit, const char* filter, int with_distance, char* out, int64_t cap) {
  std::string f(field);
  std::vector<std::string> fields;
  std::string cur;
  for (const char* p = fields_csv; *p; ++p) {
    if (*p == ',') {
      if (!cur.empty()) fields.push_back(cur);
      cur.clear();
    } else {
      cur.push_back(*p);
    }
  }
  if (!cur.empty()) fields.push_back(cur);
  vectordb::Json result, facets_cfg, facets;
  facets_cfg.LoadFromString("[]");
  vectordb::Status st;
  std::string s;
  try {
    st = static_cast<vectordb::engine::DBServer*>(h)->Search(db, table, f, fields, d, q, limit, result, filter,
                                                             with_distance != 0, facets_cfg, facets);
    s = st.ok() ? result.DumpToString() : st.message();
  } catch (const std::exception& e) {  // the gfx950 executor throws on infrastructure failures (no device)
    st = vectordb::Status(vectordb::INFRA_UNEXPECTED_ERROR, e.what());
    s = e.what();
  }
  if (cap > 0) {
    size_t n = std::min<size_t>(s.size(), (size_t)cap - 1);
    memcpy(out, s.data(), n);
    out[n] = 0;
  }
  return st.code();
}

#ifdef EPS_DROPIN
// The drop-in's C++-level batched entry (include/epsdrop/search_batch.hpp): nq vectors, ONE device batch; result = JSON array of nq
// arrays of records.  Only in the drop-in build (the reference has no batched entry to compare with: the test compares it with nq
// ref_db_search calls on both builds).
int ref_db_search_batch(void* h, const char* db, const char* table, const char* field, const char* fields_csv, float* q, int64_t nq,
                        int64_t d, int64_t limit, const char* filter, int with_distance, char* out, int64_t cap) {
  std::vector<std::string> fields;
  std::string cur;
  for (const char* p = fields_csv; *p; ++p) {
    if (*p == ',') {
      if (!cur.empty()) fields.push_back(cur);
      cur.clear();
    } else {
      cur.push_back(*p);
    }
  }
  if (!cur.empty()) fields.push_back(cur);
  vectordb::Json result;
  vectordb::Status st = epsdrop::SearchBatch(*static_cast<vectordb::engine::DBServer*>(h), db, table, field, fields, q, nq, d, limit, result, filter,
                                             with_distance != 0);
  const std::string s = st.ok() ? result.DumpToString() : st.message();
  if (cap > 0) {
    size_t n = std::min<size_t>(s.size(), (size_t)cap - 1);
    memcpy(out, s.data(), n);
    out[n] = 0;
  }
  return st.code();
}
#endif

#ifdef EPS_DROPIN
// The drop-in's bulk ingest (include/epsdrop/insert_array.hpp) for ctypes callers: `ncols` columns; names[c] = schema field, kinds[c] =
// epsdrop::ColumnView::Kind (0 i8, 1 i16, 2 i32, 3 i64, 4 u8, 5 f32, 6 f64, 7 strings), data[c] = the column's buffer (strings: n
// NUL-terminated strings laid end to end), widths[c] = 1 or the vector dimension.  out2 = {inserted, skipped}.
int ref_db_insert_array(void* h, const char* db, const char* table, int ncols, const char** names, const int* kinds, const void** data, const int64_t* widths,
                        int64_t n, int upsert, int sync, int64_t* out2, char* msg, int64_t cap) {
  std::vector<epsdrop::ColumnView> cols((size_t)ncols);
  std::vector<std::vector<std::string>> strs((size_t)ncols);
  for (int c = 0; c < ncols; ++c) {
    cols[(size_t)c].name = names[c];
    cols[(size_t)c].kind = (epsdrop::ColumnView::Kind)kinds[c];
    cols[(size_t)c].width = widths[c];
    if (kinds[c] == (int)epsdrop::ColumnView::STR) {
      const char* p = static_cast<const char*>(data[c]);
      for (int64_t i = 0; i < n; ++i) {
        strs[(size_t)c].emplace_back(p);
        p += strs[(size_t)c].back().size() + 1;
      }
      cols[(size_t)c].strings = &strs[(size_t)c];
    } else {
      cols[(size_t)c].data = data[c];
    }
  }
  epsdrop::InsertArrayResult r;
  vectordb::Status st = epsdrop::InsertArray(*static_cast<vectordb::engine::DBServer*>(h), db, table, cols, n, upsert != 0, sync != 0, &r);
  if (out2) {
    out2[0] = r.inserted;
    out2[1] = r.skipped;
  }
  if (cap > 0) {
    const std::string s = st.message();
    size_t m = std::min<size_t>(s.size(), (size_t)cap - 1);
    memcpy(msg, s.data(), m);
    msg[m] = 0;
  }
  return st.code();
}
#endif

// DBServer::Project (the "get" path: VecSearchExecutor::SearchByAttribute underneath, no vector arithmetic).
int ref_db_get(void* h, const char* db, const char* table, const char* fields_csv, const char* pk_json, const char* filter,
               int64_t skip, int64_t limit, char* out, int64_t cap) {
  std::vector<std::string> fields;
  std::string cur;
  for (const char* p = fields_csv; *p; ++p) {
    if (*p == ',') {
      if (!cur.empty()) fields.push_back(cur);
      cur.clear();
    } else {
      cur.push_back(*p);
    }
  }
  if (!cur.empty()) fields.push_back(cur);
  vectordb::Json pks, result, facets_cfg, facets;
  pks.LoadFromString(pk_json && *pk_json ? pk_json : "[]");
  facets_cfg.LoadFromString("[]");
  vectordb::Status st;
  std::string s;
  try {
    st = static_cast<vectordb::engine::DBServer*>(h)->Project(db, table, fields, pks, filter, skip, limit, result, facets_cfg, facets);
    s = st.ok() ? result.DumpToString() : st.message();
  } catch (const std::exception& e) {
    st = vectordb::Status(vectordb::INFRA_UNEXPECTED_ERROR, e.what());
    s = e.what();
  }
  if (cap > 0) {
    size_t n = std::min<size_t>(s.size(), (size_t)cap - 1);
    memcpy(out, s.data(), n);
    out[n] = 0;
  }
  return st.code();
}

// nq single-vector DBServer::Search calls issued from `threads` client threads (what concurrent REST requests do);
// first_ids[q] = "ID" of the best hit of query q (or -1).  Returns elapsed seconds, or -1 on error.
double ref_db_search_mt_filter(void* h, const char* db, const char* table, const char* field, float* queries, int64_t nq, int64_t d,
                               int64_t limit, int threads, const char* filter, int64_t* first_ids) {
  const std::string flt(filter ? filter : "");
  auto* srv = static_cast<vectordb::engine::DBServer*>(h);
  std::atomic<int64_t> next{0};
  std::atomic<int> bad{0};
  auto worker = [&]() {
    std::string f(field);
    std::vector<std::string> fields{"ID"};
    for (;;) {
      const int64_t q = next.fetch_add(1);
      if (q >= nq) break;
      vectordb::Json result, facets_cfg, facets;
      facets_cfg.LoadFromString("[]");
      try {
        auto st = srv->Search(db, table, f, fields, d, queries + q * d, limit, result, flt, true, facets_cfg, facets);
        if (!st.ok()) bad++;
        first_ids[q] = result.GetSize() > 0 ? result.GetArrayElement(0).GetInt("ID") : -1;
      } catch (const std::exception&) {
        bad++;
        first_ids[q] = -1;
      }
    }
  };
  auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) pool.emplace_back(worker);
  for (auto& t : pool) t.join();
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return bad.load() ? -1.0 : sec;
}

double ref_db_search_mt(void* h, const char* db, const char* table, const char* field, float* queries, int64_t nq, int64_t d,
                        int64_t limit, int threads, int64_t* first_ids) {
  return ref_db_search_mt_filter(h, db, table, field, queries, nq, d, limit, threads, "", first_ids);
}

int ref_omp_max_threads() { return omp_get_max_threads(); }

}  // extern "C"
