// Drop-in replacement for the reference's engine/db/ann_graph_segment.hpp (ANNGraphSegment, :22-55).
// Put this repository's include/ AHEAD of <reference>/engine on the include path: table_mvp.hpp:8,
// knn.hpp:11 and nsg.hpp:9 include "db/ann_graph_segment.hpp" by engine-relative path, so they resolve here and the
// reference's TableMVP / DBMVP / DBServer compile unmodified.  Same public members, same method signatures, same
// exceptions, same file format; BuildFromVectorTable runs on the MI355X through libepsilla_gfx950.so
// (eps_index_build) instead of NN-Descent + NsgIndex on the host.  Implementation: dropin/ann_graph_segment.cpp.
#pragma once

#include <atomic>
#include <cstdint>
#include <memory>
#include <string>
#include <unordered_map>
#include <variant>

#include "db/catalog/meta.hpp"
#include "db/vector.hpp"
#include "logger/logger.hpp"
#include "utils/concurrent_bitset.hpp"
#include "utils/concurrent_hashmap.hpp"
#include "utils/status.hpp"

namespace vectordb {
namespace engine {

using VectorColumnData = std::variant<DenseVectorColumnDataContainer,
                                      // pointer, to avoid a deep copy (same alias as the reference, :17-20)
                                      VariableLenAttrColumnContainer*>;

class ANNGraphSegment {
 public:
  explicit ANNGraphSegment(bool skip_disk_sync);
  // loads <db_catalog_path>/<table_id>/ann_graph_<field_id>.bin, or creates the directory and an empty graph file;
  // throws std::runtime_error / std::string exactly where the reference does (ann_graph_segment.cpp:53,88,95)
  explicit ANNGraphSegment(const std::string& db_catalog_path, int64_t table_id, int64_t field_id);
  explicit ANNGraphSegment(int64_t size_limit);

  void BuildFromVectorTable(VectorColumnData vector_column, int64_t n, int64_t dim, meta::MetricType metricType);
  void Debug();
  Status SaveANNGraph(const std::string& db_catalog_path, int64_t table_id, int64_t field_id, bool force = false);
  ~ANNGraphSegment();

 public:
  vectordb::engine::Logger logger_;
  bool skip_sync_disk_;
  int64_t first_record_id_;
  std::atomic<int64_t> record_number_;
  int64_t* offset_table_;    // new[]-owned CSR offsets, record_number_ + 1 entries
  int64_t* neighbor_list_;   // new[]-owned CSR neighbours
  int64_t navigation_point_;
  std::shared_ptr<void> device_mirror_;   // (additive) keeps the field's device mirror - which holds this graph - alive with the graph
  // (additive) identity of this segment for the device mirror's bookkeeping ("whose graph is in HBM"): a process-wide serial number, never
  // reused - the segment's ADDRESS is (a freed segment's address can be handed to the next one).  Travels as an opaque key.
  const uint64_t uid_ = NextUid();
  const void* OwnerKey() const { return reinterpret_cast<const void*>(static_cast<uintptr_t>(uid_)); }

 private:
  static uint64_t NextUid() {
    static std::atomic<uint64_t> next{1};
    return next.fetch_add(1, std::memory_order_relaxed);
  }
};

}  // namespace engine
}  // namespace vectordb
