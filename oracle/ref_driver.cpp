// ORACLE / TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// C-ABI facade over the *reference's own* classes, compiled together with the reference
// sources where they lie under /root/reference/engine (see oracle/Makefile) into
// oracle/_ref/libepsilla_ref.so.  Nothing here re-implements the algorithm: every entry
// point forwards to the reference (`fvec_L2sqr`, `GetDistFunc`, `ANNGraphSegment`,
// `VecSearchExecutor`, `DBServer`).  It exists so that Python tests (ctypes) can
//   (a) pin oracle/epsilla_oracle.c against the real reference on the same inputs,
//   (b) generate the golden fixtures under tests/golden/ (scripts/gen_golden.py),
//   (c) time the reference CPU path as bench.py's cpu_baseline.kind == "reference".
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
#include <omp.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <stdexcept>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "config/config.hpp"
#include "db/ann_graph_segment.hpp"
#include "db/db_server.hpp"
#include "db/execution/vec_search_executor.hpp"
#include "db/index/index.hpp"
#include "db/vector.hpp"
#include "db/index/distances.hpp"
#include "db/index/knn/knn.hpp"
#include "db/index/nsg/nsg.hpp"
#include "query/expr/expr.hpp"
#include "query/expr/expr_evaluator.hpp"
#include "utils/concurrent_bitset.hpp"

namespace vectordb { namespace engine { namespace index { extern unsigned int seed; } } }  // nsg.cpp:19

using vectordb::engine::ANNGraphSegment;
using vectordb::engine::execution::VecSearchExecutor;
namespace meta = vectordb::engine::meta;

namespace {
meta::MetricType ToMetric(int m) {
  // 0 = EUCLIDEAN, 1 = COSINE, 2 = DOT_PRODUCT  (same numbering as include/epsilla_gfx950.h)
  switch (m) {
    case 1: return meta::MetricType::COSINE;
    case 2: return meta::MetricType::DOT_PRODUCT;
    default: return meta::MetricType::EUCLIDEAN;
  }
}

struct RefExecutor {
  std::shared_ptr<ANNGraphSegment> graph;
  std::unique_ptr<VecSearchExecutor> exec;
  size_t dim;  // dist_func_param_ points here (reference passes &vector_dimension_)
};

std::atomic<uint64_t> g_dist_calls{0};
vectordb::DenseVecDistFunc<float> g_inner_fn = nullptr;
float CountingDist(const void* a, const void* b, const void* p) {
  g_dist_calls.fetch_add(1, std::memory_order_relaxed);
  return g_inner_fn(a, b, p);
}
}  // namespace

extern "C" {

// ---------------------------------------------------------------- distances
float ref_fvec_L2sqr(const float* x, const float* y, int64_t d) { return vectordb::fvec_L2sqr(x, y, (size_t)d); }
float ref_fvec_inner_product(const float* x, const float* y, int64_t d) {
  return vectordb::fvec_inner_product(x, y, (size_t)d);
}
// distance as the engine sees it: GetDistFunc(VECTOR_FLOAT, metric)(row, query, &dim)
float ref_dist(int metric, const float* row, const float* query, int64_t d) {
  auto f = std::get<vectordb::DenseVecDistFunc<float>>(vectordb::GetDistFunc(meta::FieldType::VECTOR_FLOAT, ToMetric(metric)));
  size_t dim = (size_t)d;
  return f(row, query, &dim);
}
void ref_dist_batch(int metric, const float* rows, int64_t n, const float* query, int64_t d, float* out) {
  auto f = std::get<vectordb::DenseVecDistFunc<float>>(vectordb::GetDistFunc(meta::FieldType::VECTOR_FLOAT, ToMetric(metric)));
  size_t dim = (size_t)d;
#pragma omp parallel for
  for (int64_t i = 0; i < n; ++i) out[i] = f(rows + i * d, query, &dim);
}
void ref_normalize(float* v, int64_t d) { vectordb::engine::Normalize(v, (size_t)d); }

// ---------------------------------------------------------------- graph segment
void* ref_graph_build(float* rows, int64_t n, int64_t d, int metric, int threads) {
  omp_set_num_threads(threads);  // TableMVP::Rebuild does omp_set_num_threads(RebuildThreads) (table_mvp.cpp:96)
  auto* g = new std::shared_ptr<ANNGraphSegment>(std::make_shared<ANNGraphSegment>(true));
  (*g)->BuildFromVectorTable(rows, n, d, ToMetric(metric));
  return g;
}
void* ref_graph_from_arrays(int64_t n, const int64_t* off, const int64_t* nbr, int64_t nav) {
  auto* g = new std::shared_ptr<ANNGraphSegment>(std::make_shared<ANNGraphSegment>(true));
  (*g)->record_number_ = n;
  (*g)->offset_table_ = new int64_t[n + 1];
  memcpy((*g)->offset_table_, off, sizeof(int64_t) * (n + 1));
  int64_t e = off[n];
  (*g)->neighbor_list_ = new int64_t[e > 0 ? e : 1];
  memcpy((*g)->neighbor_list_, nbr, sizeof(int64_t) * e);
  (*g)->navigation_point_ = nav;
  return g;
}
// file ctor: <dir>/<table_id>/ann_graph_<field_id>.bin
void* ref_graph_load(const char* dir, int64_t table_id, int64_t field_id) {
  try {
    return new std::shared_ptr<ANNGraphSegment>(std::make_shared<ANNGraphSegment>(std::string(dir), table_id, field_id));
  } catch (...) {
    return nullptr;
  }
}
int ref_graph_save(void* h, const char* dir, int64_t table_id, int64_t field_id) {
  auto& g = *static_cast<std::shared_ptr<ANNGraphSegment>*>(h);
  return g->SaveANNGraph(std::string(dir), table_id, field_id, true).code();
}
int64_t ref_graph_n(void* h) { return (*static_cast<std::shared_ptr<ANNGraphSegment>*>(h))->record_number_; }
int64_t ref_graph_edges(void* h) {
  auto& g = *static_cast<std::shared_ptr<ANNGraphSegment>*>(h);
  return g->offset_table_ ? g->offset_table_[g->record_number_] : 0;
}
int64_t ref_graph_nav(void* h) { return (*static_cast<std::shared_ptr<ANNGraphSegment>*>(h))->navigation_point_; }
void ref_graph_copy(void* h, int64_t* off, int64_t* nbr) {
  auto& g = *static_cast<std::shared_ptr<ANNGraphSegment>*>(h);
  int64_t n = g->record_number_;
  memcpy(off, g->offset_table_, sizeof(int64_t) * (n + 1));
  memcpy(nbr, g->neighbor_list_, sizeof(int64_t) * g->offset_table_[n]);
}
void ref_graph_free(void* h) { delete static_cast<std::shared_ptr<ANNGraphSegment>*>(h); }

// ---------------------------------------------------------------- build stages, separately
// NN-Descent kNN graph exactly as BuildFromVectorTable runs it (ann_graph_segment.cpp:208): ids per
// node in ascending distance, -1 padded to K.
void ref_knn_graph(float* rows, int64_t n, int64_t d, int64_t K, int metric, int threads, int64_t* out) {
  omp_set_num_threads(threads);
  vectordb::engine::index::Graph knng(n);
  vectordb::engine::index::KNNGraph graph(n, d, K, rows, knng, ToMetric(metric));
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = 0; j < K; ++j) out[i * K + j] = j < (int64_t)knng[i].size() ? knng[i][j] : -1;
}
// NsgIndex::Build on a caller-supplied kNN graph (always Metric_Type_L2, ann_graph_segment.cpp:216);
// the global rand_r seed (nsg.cpp:19) is reset to `seed0` first so runs are reproducible.
// Returns a graph handle (same type as ref_graph_build).
void* ref_nsg_from_knn(float* rows, int64_t n, int64_t d, const int64_t* knn, int64_t K, int64_t search_length,
                       int64_t out_degree, int64_t candidate_pool, int threads, unsigned seed0) {
  omp_set_num_threads(threads);
  vectordb::engine::index::seed = seed0;
  vectordb::engine::index::Graph knng(n);
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = 0; j < K; ++j)
      if (knn[i * K + j] >= 0) knng[i].push_back(knn[i * K + j]);
  vectordb::engine::index::BuildParams bp;
  bp.search_length = search_length;
  bp.out_degree = out_degree;
  bp.candidate_pool_size = candidate_pool;
  vectordb::engine::index::NsgIndex idx(d, n, vectordb::engine::index::NsgIndex::Metric_Type_L2);
  idx.SetKnnGraph(knng);
  idx.Build(n, rows, nullptr, bp);
  auto* g = new std::shared_ptr<ANNGraphSegment>(std::make_shared<ANNGraphSegment>(true));
  (*g)->record_number_ = n;
  (*g)->offset_table_ = new int64_t[n + 1];
  int64_t e = 0;
  for (int64_t i = 0; i < n; ++i) e += idx.nsg[i].size();
  (*g)->neighbor_list_ = new int64_t[e > 0 ? e : 1];
  int64_t o = 0;
  for (int64_t i = 0; i < n; ++i) {
    (*g)->offset_table_[i] = o;
    for (auto v : idx.nsg[i]) (*g)->neighbor_list_[o++] = v;
  }
  (*g)->offset_table_[n] = o;
  (*g)->navigation_point_ = idx.navigation_point;
  return g;
}

// The reference's own SyncPrune tail (sort + closest) and SelectEdge (nsg.cpp:557-567, 655-685, protected members) on a
// caller-supplied candidate list: pins eo_select_edge and, through it, the device's prune kernel.
namespace {
struct NsgOpen : vectordb::engine::index::NsgIndex {
  using NsgIndex::NsgIndex;
  using NsgIndex::SelectEdge;
  using NsgIndex::InterInsert;
};
}  // namespace
// NsgIndex::InterInsert (nsg.cpp:583-653) for every node in order, as Link() runs it (nsg.cpp:531-536: the omp-for is orphaned),
// on caller-supplied edge lists (ids: n * out_degree, deg[v] valid each); distances as SyncPrune stores them.
void ref_inter_insert(float* rows, int64_t n, int64_t d, const int64_t* ids, const int64_t* deg, int64_t out_degree, int64_t* out_ids,
                      int64_t* out_deg) {
  NsgOpen idx(d, n, vectordb::engine::index::NsgIndex::Metric_Type_L2);
  idx.ori_data_ = rows;
  idx.ids_ = nullptr;
  idx.ntotal = n;
  idx.out_degree = out_degree;
  idx.nsg.assign((size_t)n, {});
  std::vector<float> cut((size_t)n * out_degree);
  for (int64_t v = 0; v < n; ++v) {
    for (int64_t i = 0; i < deg[v]; ++i) {
      const int64_t u = ids[v * out_degree + i];
      idx.nsg[v].push_back((vectordb::engine::index::node_t)u);
      cut[v * out_degree + i] = idx.distance_->Compare(rows + v * d, rows + u * d, d);
    }
    if (deg[v] < out_degree) cut[v * out_degree + deg[v]] = -1;
  }
  std::vector<std::mutex> mutex_vec((size_t)n);
  for (unsigned v = 0; v < (unsigned)n; ++v) idx.InterInsert(v, mutex_vec, cut.data());
  for (int64_t v = 0; v < n; ++v) {
    out_deg[v] = (int64_t)idx.nsg[v].size();
    for (int64_t i = 0; i < out_degree; ++i) out_ids[v * out_degree + i] = i < (int64_t)idx.nsg[v].size() ? (int64_t)idx.nsg[v][i] : -1;
  }
}
int64_t ref_select_edge(float* rows, int64_t n, int64_t d, int64_t node, const int64_t* cands, int64_t m, int64_t depth, int64_t out_degree,
                        int64_t* out) {
  using vectordb::engine::index::Neighbor;
  NsgOpen idx(d, n, vectordb::engine::index::NsgIndex::Metric_Type_L2);
  idx.ori_data_ = rows;
  idx.ids_ = nullptr;   // (left uninitialised by the constructor, deleted by the destructor)
  idx.ntotal = n;
  idx.out_degree = out_degree;
  idx.candidate_pool_size = depth > 0 ? depth : 0;
  std::vector<Neighbor> pool;
  for (int64_t i = 0; i < m; ++i) {
    if (cands[i] < 0) continue;
    const float dist = idx.distance_->Compare(rows + node * d, rows + cands[i] * d, d);
    pool.emplace_back(Neighbor(cands[i], dist, true));
  }
  if (pool.empty()) return 0;
  std::stable_sort(pool.begin(), pool.end());   // (the reference's std::sort is unstable on equal distances; inputs are tie-free)
  unsigned cursor = 0;
  if (pool[cursor].id == static_cast<vectordb::engine::index::node_t>(node)) cursor++;
  if (cursor >= pool.size()) return 0;
  std::vector<Neighbor> result;
  result.push_back(pool[cursor]);
  idx.SelectEdge(cursor, pool, result, depth > 0);
  for (size_t i = 0; i < result.size(); ++i) out[i] = result[i].id;
  return (int64_t)result.size();
}

// ---------------------------------------------------------------- executor (graph search only)
void* ref_executor_new(void* graph, float* rows, int64_t d, int metric, int T, int64_t L_master, int64_t L_local,
                       int64_t iters, int count_dists) {
  auto* e = new RefExecutor;
  e->graph = *static_cast<std::shared_ptr<ANNGraphSegment>*>(graph);
  e->dim = (size_t)d;
  vectordb::DistFunc df = vectordb::GetDistFunc(meta::FieldType::VECTOR_FLOAT, ToMetric(metric));
  if (count_dists) {
    g_inner_fn = std::get<vectordb::DenseVecDistFunc<float>>(df);
    df = (vectordb::DenseVecDistFunc<float>)CountingDist;
  }
  e->exec.reset(new VecSearchExecutor(d, e->graph->navigation_point_, e->graph, e->graph->offset_table_,
                                      e->graph->neighbor_list_, rows, df, &e->dim, T, L_master, L_local, iters, false));
  return e;
}
void ref_executor_init_ids(void* h, int64_t* out) {
  auto* e = static_cast<RefExecutor*>(h);
  memcpy(out, e->exec->init_ids_.data(), sizeof(int64_t) * e->exec->init_ids_.size());
}
// VecSearchExecutor::SearchImpl on one query; copies the first K entries of the master queue.
void ref_executor_search_impl(void* h, float* query, int64_t K, int64_t* ids, float* dists) {
  auto* e = static_cast<RefExecutor*>(h);
  auto& x = *e->exec;
  x.SearchImpl(query, K, x.L_master_, x.set_L_, x.init_ids_, x.search_result_, x.L_local_, x.local_queues_starts_,
               x.local_queues_sizes_, x.is_visited_, x.subsearch_iterations_);
  const int64_t ms = x.local_queues_starts_[x.num_threads_ - 1];
  for (int64_t i = 0; i < K; ++i) {
    ids[i] = x.set_L_[ms + i].id_;
    dists[i] = x.set_L_[ms + i].distance_;
  }
}
// nq queries with E executors' worth of outer parallelism is done by the caller; this is one
// executor, sequential over queries (what one pool slot does).  Returns seconds.
double ref_executor_search_many(void* h, float* queries, int64_t nq, int64_t K, int64_t* ids, float* dists) {
  auto* e = static_cast<RefExecutor*>(h);
  auto t0 = std::chrono::steady_clock::now();
  for (int64_t q = 0; q < nq; ++q) ref_executor_search_impl(h, queries + q * (int64_t)e->dim, K, ids + q * K, dists + q * K);
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
uint64_t ref_dist_calls_reset() { return g_dist_calls.exchange(0); }
void ref_executor_free(void* h) { delete static_cast<RefExecutor*>(h); }

// ---------------------------------------------------------------- CPU baseline legs (bench.py cpu_baseline)
// Row buffer for the baseline legs: page-aligned, first-touched by the same static OpenMP schedule the reference's
// `#pragma omp parallel for` over rows uses (vec_search_executor.cpp:729), so that on a multi-socket host every
// thread later streams rows from its own NUMA node.
float* ref_alloc_rows(int64_t n, int64_t d, int threads) {
  float* p = nullptr;
  if (posix_memalign(reinterpret_cast<void**>(&p), 4096, sizeof(float) * (size_t)n * (size_t)d) != 0) return nullptr;
  omp_set_num_threads(threads);
#pragma omp parallel for
  for (int64_t i = 0; i < n; ++i) memset(p + i * d, 0, sizeof(float) * (size_t)d);
  return p;
}
void ref_free_rows(float* p) { free(p); }

// The reference's own BruteForceSearch (vec_search_executor.cpp:717-768: omp-parallel distances into Candidate[n],
// serial compaction, std::sort of all n survivors) for nq queries, one after another, with `threads` OpenMP threads.
// Writes the first k (id, dist) of every query and the per-query seconds; returns the total seconds.
double ref_bruteforce_many(float* rows, int64_t n, int64_t d, int metric, int threads, float* queries, int64_t nq, int64_t k,
                           int64_t* ids, float* dists, double* per_query_s) {
  auto g = std::make_shared<ANNGraphSegment>(true);   // record_number_ = 0 < BruteforceThreshold: no PrepareInitIds
  size_t dim = (size_t)d;
  vectordb::DistFunc df = vectordb::GetDistFunc(meta::FieldType::VECTOR_FLOAT, ToMetric(metric));
  VecSearchExecutor ex(d, 0, g, nullptr, nullptr, rows, df, &dim, threads, 500, 500, 15, false);
  vectordb::ConcurrentBitset deleted(n);
  std::vector<vectordb::query::expr::ExprNodePtr> nodes;
  std::unordered_map<std::string, size_t> offs;
  int64_t prim = 0, nvar = 0;
  std::vector<vectordb::engine::VariableLenAttrColumnContainer> var;
  vectordb::query::expr::ExprEvaluator ev(nodes, offs, prim, nvar, nullptr, var);
  double total = 0;
  for (int64_t q = 0; q < nq; ++q) {
    omp_set_num_threads(threads);
    auto t0 = std::chrono::steady_clock::now();
    ex.BruteForceSearch(queries + q * d, 0, n, deleted, ev, nullptr, -1);
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    total += sec;
    if (per_query_s) per_query_s[q] = sec;
    for (int64_t i = 0; i < k; ++i) {
      const bool ok = i < (int64_t)ex.brute_force_queue_.size();
      ids[q * k + i] = ok ? ex.brute_force_queue_[i].id_ : -1;
      dists[q * k + i] = ok ? ex.brute_force_queue_[i].distance_ : 0.f;
    }
  }
  return total;
}

// The reference's own PreFilterBruteForceSearch (vec_search_executor.cpp:770-831) with a real filter expression: `filter` is
// parsed by the reference's Expr::ParseNodeFromStr over a table with ONE INT4 attribute "ID" whose packed attribute rows are
// `id_column` (4 bytes per row), and evaluated per row by the reference's ExprEvaluator::LogicalEvaluate - BASELINE configs[3]
// ("COSINE + ID<N metadata filter") on the reference side.  Returns the total seconds; result counts in counts[q].
double ref_prefilter_many(float* rows, int64_t n, int64_t d, int metric, int threads, int32_t* id_column, const char* filter, float* queries,
                          int64_t nq, int64_t k, int64_t* ids, float* dists, int64_t* counts, double* per_query_s) {
  auto g = std::make_shared<ANNGraphSegment>(true);
  size_t dim = (size_t)d;
  vectordb::DistFunc df = vectordb::GetDistFunc(meta::FieldType::VECTOR_FLOAT, ToMetric(metric));
  VecSearchExecutor ex(d, 0, g, nullptr, nullptr, rows, df, &dim, threads, 500, 500, 15, true);
  vectordb::ConcurrentBitset deleted(n);
  std::vector<vectordb::query::expr::ExprNodePtr> nodes;
  std::unordered_map<std::string, meta::FieldType> field_map{{"ID", meta::FieldType::INT4}};
  auto st = vectordb::query::expr::Expr::ParseNodeFromStr(filter, nodes, field_map);
  if (!st.ok()) return -1.0;
  std::unordered_map<std::string, size_t> offs{{"ID", 0}};
  int64_t prim = 4, nvar = 0;
  std::vector<vectordb::engine::VariableLenAttrColumnContainer> var;
  vectordb::query::expr::ExprEvaluator ev(nodes, offs, prim, nvar, reinterpret_cast<char*>(id_column), var);
  const int root = (int)nodes.size() - 1;
  double total = 0;
  for (int64_t q = 0; q < nq; ++q) {
    omp_set_num_threads(threads);
    auto t0 = std::chrono::steady_clock::now();
    ex.PreFilterBruteForceSearch(queries + q * d, 0, n, deleted, ev, nullptr, root);
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    total += sec;
    if (per_query_s) per_query_s[q] = sec;
    if (counts) counts[q] = (int64_t)ex.brute_force_queue_.size();
    for (int64_t i = 0; i < k; ++i) {
      const bool ok = i < (int64_t)ex.brute_force_queue_.size();
      ids[q * k + i] = ok ? ex.brute_force_queue_[i].id_ : -1;
      dists[q * k + i] = ok ? ex.brute_force_queue_[i].distance_ : 0.f;
    }
  }
  return total;
}

// The reference's concurrency model for graph search (executor_pool.hpp:10-46 + SearchImpl's OpenMP team): E executors,
// each driven by its own request thread and spawning T OpenMP workers per query; queries are handed out from a shared
// counter.  Writes the first K master-queue entries and the latency of every query; returns the wall seconds.
double ref_pool_search(void* graph, float* rows, int64_t d, int metric, int E, int T, int64_t L, int64_t iters, float* queries,
                       int64_t nq, int64_t K, int64_t* ids, float* dists, double* latency_s) {
  auto& g = *static_cast<std::shared_ptr<ANNGraphSegment>*>(graph);
  std::vector<std::unique_ptr<RefExecutor>> pool;
  for (int e = 0; e < E; ++e) {
    auto r = std::make_unique<RefExecutor>();
    r->graph = g;
    r->dim = (size_t)d;
    vectordb::DistFunc df = vectordb::GetDistFunc(meta::FieldType::VECTOR_FLOAT, ToMetric(metric));
    r->exec.reset(new VecSearchExecutor(d, g->navigation_point_, g, g->offset_table_, g->neighbor_list_, rows, df, &r->dim, T, L, L,
                                        iters, false));
    pool.push_back(std::move(r));
  }
  std::atomic<int64_t> next{0};
  auto worker = [&](int e) {
    auto& x = *pool[e]->exec;
    for (;;) {
      const int64_t q = next.fetch_add(1);
      if (q >= nq) break;
      auto t0 = std::chrono::steady_clock::now();
      x.SearchImpl(queries + q * d, K, x.L_master_, x.set_L_, x.init_ids_, x.search_result_, x.L_local_, x.local_queues_starts_,
                   x.local_queues_sizes_, x.is_visited_, x.subsearch_iterations_);
      const int64_t ms = x.local_queues_starts_[x.num_threads_ - 1];
      for (int64_t i = 0; i < K; ++i) {
        ids[q * K + i] = x.set_L_[ms + i].id_;
        dists[q * K + i] = x.set_L_[ms + i].distance_;
      }
      if (latency_s) latency_s[q] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
  };
  auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (int e = 0; e < E; ++e) th.emplace_back(worker, e);
  for (auto& t : th) t.join();
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // extern "C"
// (the DBServer-level entry points - ref_db_* - are dropin/db_driver.cpp, compiled into this library over the reference's own DBServer)
