// epsdrop::SearchBatch (include/epsdrop/search_batch.hpp): the batched image of DBServer::Search -> TableMVP::Search
// (engine/db/db_server.cpp:458-510, engine/db/table_mvp.cpp:299-380), ending in ONE VecSearchExecutor::SearchBatch.
#include "epsdrop/search_batch.hpp"

#include <algorithm>
#include <mutex>
#include <stdexcept>

#include "db/table_mvp.hpp"
#include "db/vector.hpp"
#include "query/expr/expr.hpp"
#include "utils/error.hpp"

namespace epsdrop {

using vectordb::Status;
namespace meta = vectordb::engine::meta;

Status SearchBatch(vectordb::engine::DBServer& server, const std::string& db_name, const std::string& table_name, const std::string& field_name_in,
                   const float* queries, int64_t nq, int64_t dim, int64_t limit, const std::string& filter, BatchHits* out) {
  if (!out) return Status(vectordb::INVALID_PAYLOAD, "SearchBatch: null result");
  *out = BatchHits();
  if (nq < 0 || (nq > 0 && !queries)) return Status(vectordb::INVALID_PAYLOAD, "SearchBatch: bad query matrix");
  // ---- DBServer::Search (:470-509)
  auto db = server.GetDB(db_name);
  if (db == nullptr) return Status(vectordb::DB_UNEXPECTED_ERROR, "DB not found: " + db_name);
  auto table = db->GetTable(table_name);
  if (table == nullptr) return Status(vectordb::DB_UNEXPECTED_ERROR, "Table not found: " + table_name);
  std::string field_name = field_name_in;
  if (field_name.empty()) {
    for (auto& field : table->table_schema_.fields_) {
      if (field.field_type_ == meta::FieldType::VECTOR_FLOAT || field.field_type_ == meta::FieldType::VECTOR_DOUBLE ||
          field.field_type_ == meta::FieldType::SPARSE_VECTOR_FLOAT || field.field_type_ == meta::FieldType::SPARSE_VECTOR_DOUBLE) {
        if (!field_name.empty()) return Status(vectordb::INVALID_PAYLOAD, "Must specify queryField if there are more than 1 vector fields.");
        field_name = field.name_;
      }
    }
  }
  std::vector<vectordb::query::expr::ExprNodePtr> filter_nodes;
  Status st = vectordb::query::expr::Expr::ParseNodeFromStr(filter, filter_nodes, table->field_name_field_type_map_);
  if (!st.ok()) return st;
  // ---- TableMVP::Search (:306-362)
  if (table->field_name_field_type_map_.find(field_name) == table->field_name_field_type_map_.end())
    return Status(vectordb::DB_UNEXPECTED_ERROR, "Field name not found: " + field_name);
  const auto field_type = table->field_name_field_type_map_[field_name];
  if (field_type != meta::FieldType::VECTOR_FLOAT && field_type != meta::FieldType::VECTOR_DOUBLE)
    return Status(vectordb::USER_ERROR, field_type == meta::FieldType::SPARSE_VECTOR_FLOAT || field_type == meta::FieldType::SPARSE_VECTOR_DOUBLE
                                            ? "SearchBatch: the query field must be a dense vector field" : "Field type is not vector.");
  std::vector<float> normalized;
  if (table->field_name_metric_type_map_[field_name] == meta::MetricType::COSINE && nq > 0) {   // (:333-343: every query is normalised)
    normalized.assign(queries, queries + (size_t)nq * dim);
    for (int64_t q = 0; q < nq; ++q) vectordb::engine::Normalize((vectordb::engine::DenseVectorPtr)(normalized.data() + (size_t)q * dim), dim);
    queries = normalized.data();
  }
  out->table = table;
  out->field = field_name;
  out->counts.assign((size_t)nq, 0);
  try {
    const int64_t field_offset = table->table_segment_->vec_field_name_executor_pool_idx_map_[field_name];
    std::unique_lock<std::mutex> lock(table->executor_pool_mutex_);   // (:359-362: pool looked up and an executor taken under the mutex)
    auto pool = table->executor_pool_.at(field_offset);
    auto executor = vectordb::engine::execution::RAIIVecSearchExecutor(pool, pool->acquire());
    lock.unlock();
    if (nq > 0 && dim != executor.exec_->dimension_) return Status(vectordb::DB_UNEXPECTED_ERROR, "Query dimension doesn't match the vector field dimension.");
    if (nq > 0)
      executor.exec_->SearchBatch(queries, nq, table->table_segment_.get(), (size_t)std::max<int64_t>(limit, 0), filter_nodes, out->ids, out->dist, out->counts,
                                  out->width);
  } catch (const std::exception& e) {   // the device executor throws on infrastructure failures (no device, out of HBM)
    return Status(vectordb::INFRA_UNEXPECTED_ERROR, e.what());
  }
  for (auto& c : out->counts) c = (int32_t)std::max<int64_t>(0, std::min<int64_t>(c, limit));   // (:374: result_num capped by limit)
  return Status::OK();
}

Status SearchBatch(vectordb::engine::DBServer& server, const std::string& db_name, const std::string& table_name, const std::string& field_name,
                   std::vector<std::string>& response_fields, const float* queries, int64_t nq, int64_t dim, int64_t limit, vectordb::Json& result,
                   const std::string& filter, bool with_distance) {
  result.LoadFromString("[]");
  BatchHits hits;
  {
    // (TableMVP::Search checks the response fields before it searches, :310-314)
    auto db = server.GetDB(db_name);
    auto table = db ? db->GetTable(table_name) : nullptr;
    if (table)
      for (auto& f : response_fields)
        if (table->field_name_field_type_map_.find(f) == table->field_name_field_type_map_.end())
          return Status(vectordb::DB_UNEXPECTED_ERROR, "Field name not found: " + f);
  }
  Status st = SearchBatch(server, db_name, table_name, field_name, queries, nq, dim, limit, filter, &hits);
  if (!st.ok()) return st;
  for (int64_t q = 0; q < nq; ++q) {
    const int64_t cnt = hits.counts[(size_t)q];
    std::vector<int64_t> ids(hits.ids.begin() + (size_t)q * hits.width, hits.ids.begin() + (size_t)q * hits.width + cnt);
    std::vector<double> dist(hits.dist.begin() + (size_t)q * hits.width, hits.dist.begin() + (size_t)q * hits.width + cnt);
    vectordb::Json rows;
    st = hits.table->Project(response_fields, cnt, ids, rows, with_distance, dist);   // (:378-380)
    if (!st.ok()) return st;
    result.AddObjectToArray(rows);
  }
  return Status::OK();
}

}  // namespace epsdrop
