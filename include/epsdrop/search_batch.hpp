// C++-level batched query entry of the drop-in.  The reference's DBServer answers ONE vector per call
// (engine/db/db_server.cpp:458-510 -> TableMVP::Search, engine/db/table_mvp.cpp:299-380; the REST handler above it,
// server/web_server/web_controller.hpp:747-761, and the Python binding, bindings/python/interface.cpp:260-331, likewise), and every
// call takes one executor from the field's pool (executor_pool.hpp:10-46).  These free functions are what a host with a BATCH of
// query vectors calls instead - the REST handler of a batched endpoint, the `epsilla.query_batch` method of the drop-in module
// (dropin/epsilla_module.cpp calls exactly this), any C++ embedding of DBServer: the same lookups, validation, COSINE
// normalisation, filter parsing and result rules as DBServer::Search, then ONE VecSearchExecutor::SearchBatch = one
// eps_index_search(nq = N) on the device.  Additive: nothing of the reference's classes changes.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "db/db_server.hpp"
#include "utils/json.hpp"
#include "utils/status.hpp"

namespace epsdrop {

struct BatchHits {
  std::vector<int64_t> ids;      // [nq][width] internal row ids, -1 beyond counts[q]
  std::vector<float> dist;       // [nq][width] distances as DBServer::Search reports them (squared L2 / 1 - cos / -dot)
  std::vector<int32_t> counts;   // [nq] results of query q (<= min(limit, width))
  int32_t width = 0;
  std::shared_ptr<vectordb::engine::TableMVP> table;   // the table the ids refer to (keeps it alive for a projection)
  std::string field;             // the vector field that was searched (resolved when the caller passed "")
};

// queries: row-major float[nq][dim].  field_name "": the table's only vector field (error if there are several), as DBServer::Search.
// Errors come back as the reference's Status (DB_NOT_FOUND, TABLE_NOT_FOUND, INVALID_EXPR, ...); device failures as
// INFRA_UNEXPECTED_ERROR (the executor throws, DBServer::Search would let that escape: here it is caught).
vectordb::Status SearchBatch(vectordb::engine::DBServer& server, const std::string& db_name, const std::string& table_name, const std::string& field_name,
                             const float* queries, int64_t nq, int64_t dim, int64_t limit, const std::string& filter, BatchHits* out);

// ... and projected, the batched image of DBServer::Search's `result`: a JSON array of nq arrays of records
// (`response_fields` + "@distance" when with_distance), TableMVP::Project per query.
vectordb::Status SearchBatch(vectordb::engine::DBServer& server, const std::string& db_name, const std::string& table_name, const std::string& field_name,
                             std::vector<std::string>& response_fields, const float* queries, int64_t nq, int64_t dim, int64_t limit,
                             vectordb::Json& result, const std::string& filter, bool with_distance);

}  // namespace epsdrop
