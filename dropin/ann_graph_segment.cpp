// Implementation of the drop-in ANNGraphSegment (include/db/ann_graph_segment.hpp) over the C ABI.
// Reference behaviour followed: engine/db/ann_graph_segment.cpp:31-266.
#include "db/ann_graph_segment.hpp"

#include <unistd.h>

#include <cstdio>
#include <fstream>
#include <stdexcept>

#include "dist_func.hpp"
#include "epsilla_gfx950.h"
#include "utils/common_util.hpp"

namespace vectordb {
namespace engine {

namespace {
std::string GraphPath(const std::string& dir, int64_t table_id, int64_t field_id) {
  return dir + "/" + std::to_string(table_id) + "/ann_graph_" + std::to_string(field_id) + ".bin";
}
int ToEpsMetric(meta::MetricType m) {
  switch (m) {
    case meta::MetricType::COSINE: return EPS_METRIC_COSINE;
    case meta::MetricType::DOT_PRODUCT: return EPS_METRIC_DOT_PRODUCT;
    default: return EPS_METRIC_EUCLIDEAN;
  }
}
}  // namespace

ANNGraphSegment::ANNGraphSegment(bool skip_sync_disk)
    : skip_sync_disk_(skip_sync_disk), first_record_id_(0), record_number_(0), offset_table_(nullptr),
      neighbor_list_(nullptr), navigation_point_(0) {}

ANNGraphSegment::ANNGraphSegment(int64_t)
    : skip_sync_disk_(true), first_record_id_(0), record_number_(0), offset_table_(nullptr), neighbor_list_(nullptr),
      navigation_point_(0) {}

ANNGraphSegment::ANNGraphSegment(const std::string& db_catalog_path, int64_t table_id, int64_t field_id)
    : skip_sync_disk_(false), first_record_id_(0), record_number_(0), offset_table_(nullptr), neighbor_list_(nullptr),
      navigation_point_(0) {
  const std::string file_path = GraphPath(db_catalog_path, table_id, field_id);
  if (server::CommonUtil::IsFileExist(file_path)) {
    std::ifstream file(file_path, std::ios::binary);
    if (!file) throw std::runtime_error("Cannot open file: " + file_path);
    int64_t n = 0;
    file.read(reinterpret_cast<char*>(&n), sizeof(n));
    file.read(reinterpret_cast<char*>(&first_record_id_), sizeof(first_record_id_));
    record_number_ = n;
    offset_table_ = new int64_t[n + 1];
    file.read(reinterpret_cast<char*>(offset_table_), sizeof(int64_t) * (n + 1));
    const int64_t edges = offset_table_[n];
    neighbor_list_ = new int64_t[edges > 0 ? edges : 1];
    file.read(reinterpret_cast<char*>(neighbor_list_), sizeof(int64_t) * edges);
    file.read(reinterpret_cast<char*>(&navigation_point_), sizeof(navigation_point_));
    // n nodes without a single edge is the placeholder a hash-sharded mirror leaves (its shards keep the real graphs, see
    // BuildFromVectorTable): nothing here can be walked, so the segment counts as "not built" and its executors scan exactly
    if (n >= 2 && edges == 0) record_number_ = 0;
  } else {
    auto mkdir_status = server::CommonUtil::CreateDirectory(db_catalog_path + "/" + std::to_string(table_id));
    if (!mkdir_status.ok()) throw mkdir_status.message();
    offset_table_ = new int64_t[1];
    offset_table_[0] = 0;
    neighbor_list_ = new int64_t[1];
    auto status = SaveANNGraph(db_catalog_path, table_id, field_id);
    if (!status.ok()) throw status.message();
  }
}

Status ANNGraphSegment::SaveANNGraph(const std::string& db_catalog_path, int64_t table_id, int64_t field_id, bool force) {
  if (skip_sync_disk_ && !force) return Status::OK();
  const std::string path = GraphPath(db_catalog_path, table_id, field_id);
  const std::string tmp_path = path + ".tmp";
  FILE* file = fopen(tmp_path.c_str(), "wb");
  if (!file) return Status(DB_UNEXPECTED_ERROR, "Cannot open file: " + path);
  const int64_t n = record_number_;
  fwrite(&n, sizeof(n), 1, file);
  fwrite(&first_record_id_, sizeof(first_record_id_), 1, file);
  fwrite(offset_table_, sizeof(int64_t), n + 1, file);
  fwrite(neighbor_list_, sizeof(int64_t), offset_table_[n], file);
  fwrite(&navigation_point_, sizeof(navigation_point_), 1, file);
  fflush(file);
  fsync(fileno(file));
  fclose(file);
  if (std::rename(tmp_path.c_str(), path.c_str()) != 0)
    return Status(INFRA_UNEXPECTED_ERROR, "Failed to rename temp file: " + tmp_path + " to " + path);
  return Status::OK();
}

void ANNGraphSegment::BuildFromVectorTable(VectorColumnData vector_column, int64_t n, int64_t dim, meta::MetricType metricType) {
  if (!std::holds_alternative<DenseVectorColumnDataContainer>(vector_column))
    throw std::runtime_error("sparse-vector graphs are not built on the device (host DBMS path, SURVEY 2 row 14)");
  logger_.Debug("gfx950 graph build start");
  // The graph is built on the field's device mirror - the HBM copy of the column its executors search (r2 created a second index on
  // device 0, uploaded the whole table again and freed it).  With EPS_DEVICES the mirror is hash-sharded and every shard builds
  // and keeps the graph of its own rows on its own device.
  int64_t *off = nullptr, *nbr = nullptr, nav = 0;
  const std::string err = epsdrop::BuildGraphOnMirror(std::get<DenseVectorColumnDataContainer>(vector_column), n, dim, ToEpsMetric(metricType), OwnerKey(), &off, &nbr,
                                                      &nav, &device_mirror_);
  if (!err.empty()) throw std::runtime_error("gfx950 graph build: " + err);
  delete[] offset_table_;
  delete[] neighbor_list_;
  offset_table_ = off;
  neighbor_list_ = nbr;
  navigation_point_ = nav;
  record_number_ = n;
  logger_.Debug("gfx950 graph build finish");
}

void ANNGraphSegment::Debug() {}

ANNGraphSegment::~ANNGraphSegment() {
  delete[] offset_table_;
  delete[] neighbor_list_;
}

}  // namespace engine
}  // namespace vectordb
