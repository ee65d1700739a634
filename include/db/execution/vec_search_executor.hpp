// Drop-in replacement for the reference's engine/db/execution/vec_search_executor.hpp (VecSearchExecutor, :30-204).
// Resolved ahead of the reference's header through -I order (table_mvp.hpp:11, executor_pool.hpp:5 include it by
// engine-relative path).  Constructor signature, Search / SearchByAttribute signatures and the public result members
// TableMVP reads (search_result_, distance_, dimension_; table_mvp.cpp:365-452) are the reference's; the work runs on
// the MI355X through libepsilla_gfx950.so (eps_index_search).  Implementation: dropin/vec_search_executor.cpp.
//
// All executors of one vector field share ONE device mirror of the field (row store, graph, scratch): the
// reference's ExecutorPool creates NumExecutorPerField (16) executors per field and they must not each upload the
// table.  Calls on the shared mirror are serialised; rows appended after construction are uploaded incrementally,
// the deleted bitset is re-read on every call (as the reference re-reads table_segment->deleted_).
#pragma once

// the reference's header pulls these in and other reference files rely on that transitively
// (e.g. db/wal/write_ahead_log.hpp uses std::ifstream without including <fstream>)
#include <algorithm>
#include <cfloat>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <set>
#include <unordered_map>
#include <unordered_set>
#include <variant>
#include <vector>

#include "db/ann_graph_segment.hpp"
#include "db/catalog/meta.hpp"
#include "db/index/index.hpp"
#include "db/table_segment_mvp.hpp"
#include "db/vector.hpp"
#include "query/expr/expr_evaluator.hpp"
#include "query/expr/expr_types.hpp"
#include "utils/json.hpp"
#include "utils/status.hpp"

struct eps_filter_op;   // include/epsilla_gfx950.h

namespace vectordb {
namespace engine {
namespace execution {

constexpr const int BruteforceThreshold = 512;  // reference :28

struct DeviceField;  // shared per-field GPU mirror (dropin/vec_search_executor.cpp)
struct Pending;      // one request of the micro-batcher (same file)

class VecSearchExecutor {
 public:
  std::shared_ptr<ANNGraphSegment> ann_index_;  // keeps the graph alive while old executors drain after a rebuild
  int64_t total_indexed_vector_ = 0;
  int64_t dimension_ = 0;
  int64_t start_search_point_ = 0;
  int64_t* offset_table_;
  int64_t* neighbor_list_;
  VectorColumnData vector_column_;
  DistFunc fstdistfunc_;
  void* dist_func_param_;
  int num_threads_;
  int64_t L_master_;
  int64_t L_local_;
  int64_t subsearch_iterations_;
  bool prefilter_enabled_;
  std::vector<int64_t> search_result_;
  std::vector<double> distance_;
  bool brute_force_search_;

  VecSearchExecutor(const int64_t dimension, const int64_t start_search_point, std::shared_ptr<ANNGraphSegment> ann_index,
                    int64_t* offset_table, int64_t* neighbor_list,
                    std::variant<DenseVectorColumnDataContainer, VariableLenAttrColumnContainer*> vector_column,
                    DistFunc fstdistfunc, void* dist_func_param, int num_threads, int64_t L_master, int64_t L_local,
                    int64_t subsearch_iterations, bool prefilter_enabled);
  ~VecSearchExecutor();

  Status Search(const VectorPtr query_data, vectordb::engine::TableSegmentMVP* table_segment, const size_t limit,
                std::vector<vectordb::query::expr::ExprNodePtr>& filter_nodes, int64_t& result_size);

  Status SearchByAttribute(meta::TableSchema& table_schema, vectordb::engine::TableSegmentMVP* table_segment,
                           const size_t skip, const size_t limit, vectordb::Json& primary_keys,
                           std::vector<vectordb::query::expr::ExprNodePtr>& filter_nodes, int64_t& result_size);

  // Additive (the reference takes one vector per call): nq row-major queries answered by ONE device batch when
  // the filter is empty or compiles to a device program, with the mode selection and result caps of nq Search() calls.
  // ids / dist: [nq][width] (-1 / +inf beyond counts[q]).  Throws like Search().
  Status SearchBatch(const float* queries, int64_t nq, vectordb::engine::TableSegmentMVP* table_segment, const size_t limit,
                     std::vector<vectordb::query::expr::ExprNodePtr>& filter_nodes, std::vector<int64_t>& ids, std::vector<float>& dist,
                     std::vector<int32_t>& counts, int32_t& width);

 private:
  void FillKey(Pending& key, vectordb::engine::TableSegmentMVP* table_segment, size_t limit, const std::vector<eps_filter_op>* program);
  // unfiltered queries and queries with a device-compiled filter: coalesced with the concurrent calls of the pool's other
  // executors that carry the same filter program into one device batch
  Status SearchBatched(const float* query, vectordb::engine::TableSegmentMVP* table_segment, size_t limit, int64_t& result_size,
                       const std::vector<eps_filter_op>* program);
  std::shared_ptr<DeviceField> dev_;
  int metric_ = 0;
};

}  // namespace execution
}  // namespace engine
}  // namespace vectordb
