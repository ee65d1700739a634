// Implementation of the drop-in VecSearchExecutor (include/db/execution/vec_search_executor.hpp) over the C ABI.
// Reference behaviour followed: engine/db/execution/vec_search_executor.cpp:29-73 (ctor), :833-935 (Search),
// :937-1033 (SearchByAttribute).  The distance / traversal / top-k work is eps_index_search on the MI355X; what
// stays here is glue that needs the DBMS's own types: incremental upload of appended rows, the deleted bitset, and
// lowering of the parsed filter (ExprNode array) to either the device's `int column <op> const` form or — for
// anything else (strings, LIKE, IN, AND/OR trees, geo) — a host-evaluated visibility bitset handed to the device,
// which applies it exactly where the reference applies deleted_/LogicalEvaluate.
#include "db/execution/vec_search_executor.hpp"

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <stdexcept>
#include <tuple>

#include "dist_func.hpp"
#include "epsilla_gfx950.h"

namespace vectordb {
namespace engine {
namespace execution {

using vectordb::query::expr::ExprEvaluator;
using vectordb::query::expr::ExprNodePtr;
using vectordb::query::expr::NodeType;

// One in-flight Search() call waiting to be served by the micro-batcher.
struct Pending {
  const float* query;
  vectordb::engine::TableSegmentMVP* segment;
  int32_t k;
  // executor state the batch must agree on
  const void* graph_owner;
  int64_t graph_n, start_point;
  int64_t* off;
  int64_t* nbr;
  int T;
  int64_t L, Lq, I;
  bool prefilter;
  // results
  std::vector<int64_t> ids;
  std::vector<float> dist;
  int32_t count = 0;
  bool done = false;
  std::string error;
};

struct DeviceField {
  std::mutex mu;
  // micro-batcher: concurrent Search() calls of the pool's executors are coalesced into one
  // eps_index_search with nq > 1 by whichever caller finds the device idle
  std::mutex qmu;
  std::condition_variable qcv;
  std::deque<Pending*> queue;
  bool busy = false;
  eps_index* h = nullptr;
  const float* column = nullptr;
  int64_t dim = 0;
  int metric = 0;
  int64_t attached = 0;            // rows already in HBM
  const void* graph_owner = nullptr;  // ANNGraphSegment whose CSR is on the device
  int64_t graph_n = -1;
  std::vector<uint8_t> mask;       // scratch: deleted | !filter, for filters evaluated on the host
  ~DeviceField() {
    if (h) eps_index_destroy(h);
  }
};

namespace {
std::mutex g_mu;
std::map<std::tuple<const float*, int64_t, int>, std::weak_ptr<DeviceField>> g_fields;

std::shared_ptr<DeviceField> AcquireField(const float* column, int64_t dim, int metric) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto key = std::make_tuple(column, dim, metric);
  auto it = g_fields.find(key);
  if (it != g_fields.end()) {
    if (auto sp = it->second.lock()) return sp;
  }
  auto sp = std::make_shared<DeviceField>();
  sp->column = column;
  sp->dim = dim;
  sp->metric = metric;
  if (eps_index_create(dim, metric, 0, &sp->h) != EPS_OK) return nullptr;  // no gfx950 device: Search() reports it
  g_fields[key] = sp;
  return sp;
}

bool IsIntAttr(NodeType t) { return t == NodeType::Int1Attr || t == NodeType::Int2Attr || t == NodeType::Int4Attr || t == NodeType::Int8Attr; }
int IntWidth(NodeType t) { return t == NodeType::Int1Attr ? 1 : t == NodeType::Int2Attr ? 2 : t == NodeType::Int4Attr ? 4 : 8; }
int CmpOp(NodeType t, bool flipped) {
  switch (t) {
    case NodeType::LT: return flipped ? EPS_OP_GT : EPS_OP_LT;
    case NodeType::LTE: return flipped ? EPS_OP_GE : EPS_OP_LE;
    case NodeType::GT: return flipped ? EPS_OP_LT : EPS_OP_GT;
    case NodeType::GTE: return flipped ? EPS_OP_LE : EPS_OP_GE;
    case NodeType::EQ: return EPS_OP_EQ;
    case NodeType::NE: return EPS_OP_NE;
    default: return EPS_OP_NONE;
  }
}
bool UsesDistance(const std::vector<ExprNodePtr>& nodes) {
  for (auto& n : nodes)
    if (n && n->field_name == "@distance") return true;
  return false;
}
}  // namespace

VecSearchExecutor::VecSearchExecutor(const int64_t dimension, const int64_t start_search_point,
                                     std::shared_ptr<ANNGraphSegment> ann_index, int64_t* offset_table,
                                     int64_t* neighbor_list,
                                     std::variant<DenseVectorColumnDataContainer, VariableLenAttrColumnContainer*> vector_column,
                                     DistFunc fstdistfunc, void* dist_func_param, int num_threads, int64_t L_master,
                                     int64_t L_local, int64_t subsearch_iterations, bool prefilter_enabled)
    : ann_index_(ann_index),
      total_indexed_vector_(ann_index->record_number_),
      dimension_(dimension),
      start_search_point_(start_search_point),
      offset_table_(offset_table),
      neighbor_list_(neighbor_list),
      vector_column_(vector_column),
      fstdistfunc_(fstdistfunc),
      dist_func_param_(dist_func_param),
      num_threads_(num_threads),
      L_master_(L_master),
      L_local_(L_local),
      subsearch_iterations_(subsearch_iterations),
      prefilter_enabled_(prefilter_enabled),
      search_result_(L_master),
      distance_(L_master),
      brute_force_search_(ann_index->record_number_ < BruteforceThreshold) {
  if (std::holds_alternative<DenseVectorColumnDataContainer>(vector_column_) &&
      std::holds_alternative<DenseVecDistFunc<float>>(fstdistfunc_)) {
    metric_ = epsdrop::MetricOfDistFunc(reinterpret_cast<const void*>(std::get<DenseVecDistFunc<float>>(fstdistfunc_)));
    if (metric_ >= 0) dev_ = AcquireField(std::get<DenseVectorColumnDataContainer>(vector_column_), dimension_, metric_);
  }
}

VecSearchExecutor::~VecSearchExecutor() {}

Status VecSearchExecutor::Search(const VectorPtr query_data, vectordb::engine::TableSegmentMVP* table_segment,
                                 const size_t limit, std::vector<ExprNodePtr>& filter_nodes, int64_t& result_size) {
  result_size = 0;
  if (!std::holds_alternative<DenseVectorPtr>(query_data) || !std::holds_alternative<DenseVectorColumnDataContainer>(vector_column_))
    return Status(NOT_IMPLEMENTED_ERROR, "sparse-vector search is not served by the gfx950 executor");
  // TableMVP::Search discards the Status we return (table_mvp.cpp:372), so an infrastructure failure must not
  // look like "0 results": throw, as the constructors of the reference's own classes do on I/O failure.
  if (!dev_) throw std::runtime_error("no usable gfx950 device (libepsilla_gfx950 has no CPU fallback)");
  if (limit == 0) return Status::OK();
  DeviceField& dev = *dev_;
  {
    const int root0 = static_cast<int>(filter_nodes.size()) - 1;
    const bool unfiltered = root0 < 0 || (filter_nodes[root0]->node_type == NodeType::BoolConst && filter_nodes[root0]->bool_value);
    static const bool batching = !(getenv("EPS_DROPIN_BATCH") && atoi(getenv("EPS_DROPIN_BATCH")) == 0);
    if (unfiltered && batching) return SearchBatched(std::get<DenseVectorPtr>(query_data), table_segment, limit, result_size);
  }
  std::lock_guard<std::mutex> lk(dev.mu);
  auto fail = [&](const char* what) -> Status {
    throw std::runtime_error(std::string("gfx950 executor: ") + what + ": " + eps_index_last_error(dev.h));
  };

  // rows [0, record_number_) are immutable once written (SURVEY §8b "Ownership"): upload only the new tail
  const int64_t total_vector = table_segment->record_number_;
  if (total_vector > dev.attached) {
    const float* base = std::get<DenseVectorColumnDataContainer>(vector_column_);
    const int32_t rc = dev.attached == 0 ? eps_index_attach_rows(dev.h, base, total_vector)
                                         : eps_index_append_rows(dev.h, base + dev.attached * dimension_, total_vector - dev.attached);
    if (rc != EPS_OK) return fail("row upload");
    dev.attached = total_vector;
  }
  if (dev.graph_owner != ann_index_.get() || dev.graph_n != total_indexed_vector_) {
    if (eps_index_set_graph(dev.h, total_indexed_vector_, offset_table_, neighbor_list_, start_search_point_) != EPS_OK)
      return fail("graph upload");
    dev.graph_owner = ann_index_.get();
    dev.graph_n = total_indexed_vector_;
  }

  // ---- filter lowering
  ConcurrentBitset& deleted = *(table_segment->deleted_);
  const int root = static_cast<int>(filter_nodes.size()) - 1;
  bool device_filter = false, host_mask = false;
  if (root >= 0) {
    const ExprNodePtr& r = filter_nodes[root];
    if (r->node_type == NodeType::BoolConst && r->bool_value) {
      // always true
    } else {
      const int op0 = CmpOp(r->node_type, false);
      if (op0 != EPS_OP_NONE && r->left < filter_nodes.size() && r->right < filter_nodes.size()) {
        const ExprNodePtr& a = filter_nodes[r->left];
        const ExprNodePtr& b = filter_nodes[r->right];
        const ExprNodePtr* attr = nullptr;
        const ExprNodePtr* cst = nullptr;
        bool flipped = false;
        if (IsIntAttr(a->node_type) && b->node_type == NodeType::IntConst) {
          attr = &a;
          cst = &b;
        } else if (IsIntAttr(b->node_type) && a->node_type == NodeType::IntConst) {
          attr = &b;
          cst = &a;
          flipped = true;
        }
        if (attr) {
          const auto off = table_segment->field_name_mem_offset_map_.find((*attr)->field_name);
          if (off != table_segment->field_name_mem_offset_map_.end()) {
            if (eps_index_set_int_filter(dev.h, table_segment->attribute_table_ + off->second, table_segment->primitive_offset_,
                                         IntWidth((*attr)->node_type), CmpOp(r->node_type, flipped), (*cst)->int_value) != EPS_OK)
              return fail("filter upload");
            device_filter = true;
          }
        }
      }
      if (!device_filter) host_mask = true;
    }
  }
  if (!device_filter && eps_index_set_int_filter(dev.h, nullptr, 0, 0, EPS_OP_NONE, 0) != EPS_OK) return fail("filter reset");
  if (host_mask) {
    if (UsesDistance(filter_nodes))
      return Status(NOT_IMPLEMENTED_ERROR, "@distance filters are not lowered to the gfx950 executor yet");
    ExprEvaluator ev(filter_nodes, table_segment->field_name_mem_offset_map_, table_segment->primitive_offset_,
                     table_segment->var_len_attr_num_, table_segment->attribute_table_, table_segment->var_len_attr_table_);
    dev.mask.assign((size_t)(total_vector + 7) / 8, 0);
    for (int64_t id = 0; id < total_vector; ++id)
      if (deleted.test(id) || !ev.LogicalEvaluate(root, id)) dev.mask[id >> 3] |= uint8_t(1u << (id & 7));
    if (eps_index_set_deleted(dev.h, dev.mask.data(), (int64_t)dev.mask.size()) != EPS_OK) return fail("mask upload");
  } else {
    if (eps_index_set_deleted(dev.h, deleted.data(), (int64_t)deleted.size()) != EPS_OK) return fail("deleted upload");
  }

  eps_search_params p;
  eps_default_search_params(&p);
  p.mode = EPS_MODE_REFERENCE;
  p.prefilter = prefilter_enabled_ ? 1 : 0;
  p.intra_threads = num_threads_;
  p.master_queue = L_master_;
  p.local_queue = L_local_;
  p.sync_interval = subsearch_iterations_;
  const int32_t k = (int32_t)std::min<size_t>(limit, 1024);
  std::vector<int64_t> ids((size_t)k);
  std::vector<float> dist((size_t)k);
  int32_t count = 0;
  if (eps_index_search(dev.h, std::get<DenseVectorPtr>(query_data), 1, k, &p, ids.data(), dist.data(), &count) != EPS_OK)
    return fail("search");
  if ((size_t)count > search_result_.size()) {
    search_result_.resize(count);
    distance_.resize(count);
  }
  for (int32_t i = 0; i < count; ++i) {
    search_result_[i] = ids[i];
    distance_[i] = dist[i];
  }
  result_size = count;
  return Status::OK();
}


// ---- micro-batched path for unfiltered queries -------------------------------------------------------------------
namespace {
bool SameKey(const Pending& a, const Pending& b) {
  return a.segment == b.segment && a.k == b.k && a.graph_owner == b.graph_owner && a.graph_n == b.graph_n && a.T == b.T &&
         a.L == b.L && a.Lq == b.Lq && a.I == b.I && a.prefilter == b.prefilter;
}

void RunBatch(DeviceField& dev, int64_t dim, std::vector<Pending*>& batch) {
  std::lock_guard<std::mutex> lk(dev.mu);
  Pending& h = *batch[0];
  std::string err;
  auto fail = [&](const char* what) { err = std::string("gfx950 executor: ") + what + ": " + eps_index_last_error(dev.h); };
  const int64_t total_vector = h.segment->record_number_;
  if (total_vector > dev.attached) {
    const int32_t rc = dev.attached == 0 ? eps_index_attach_rows(dev.h, dev.column, total_vector)
                                         : eps_index_append_rows(dev.h, dev.column + dev.attached * dim, total_vector - dev.attached);
    if (rc != EPS_OK) fail("row upload"); else dev.attached = total_vector;
  }
  if (err.empty() && (dev.graph_owner != h.graph_owner || dev.graph_n != h.graph_n)) {
    if (eps_index_set_graph(dev.h, h.graph_n, h.off, h.nbr, h.start_point) != EPS_OK) fail("graph upload");
    else { dev.graph_owner = h.graph_owner; dev.graph_n = h.graph_n; }
  }
  ConcurrentBitset& deleted = *(h.segment->deleted_);
  if (err.empty() && eps_index_set_int_filter(dev.h, nullptr, 0, 0, EPS_OP_NONE, 0) != EPS_OK) fail("filter reset");
  if (err.empty() && eps_index_set_deleted(dev.h, deleted.data(), (int64_t)deleted.size()) != EPS_OK) fail("deleted upload");
  const int64_t nq = (int64_t)batch.size();
  const int32_t k = h.k;
  std::vector<float> q((size_t)nq * dim);
  std::vector<int64_t> ids((size_t)nq * k);
  std::vector<float> dist((size_t)nq * k);
  std::vector<int32_t> cnt((size_t)nq);
  for (int64_t i = 0; i < nq; ++i) std::memcpy(&q[(size_t)i * dim], batch[i]->query, sizeof(float) * dim);
  if (err.empty()) {
    eps_search_params p;
    eps_default_search_params(&p);
    p.mode = EPS_MODE_REFERENCE;
    p.prefilter = h.prefilter ? 1 : 0;
    p.intra_threads = h.T;
    p.master_queue = h.L;
    p.local_queue = h.Lq;
    p.sync_interval = h.I;
    if (eps_index_search(dev.h, q.data(), nq, k, &p, ids.data(), dist.data(), cnt.data()) != EPS_OK) fail("search");
  }
  for (int64_t i = 0; i < nq; ++i) {
    Pending& r = *batch[i];
    r.error = err;
    if (err.empty()) {
      r.count = cnt[i];
      r.ids.assign(ids.begin() + i * k, ids.begin() + i * k + cnt[i]);
      r.dist.assign(dist.begin() + i * k, dist.begin() + i * k + cnt[i]);
    }
  }
}
}  // namespace

Status VecSearchExecutor::SearchBatched(const float* query, vectordb::engine::TableSegmentMVP* table_segment, size_t limit,
                                        int64_t& result_size) {
  DeviceField& dev = *dev_;
  Pending me;
  me.query = query;
  me.segment = table_segment;
  me.k = (int32_t)std::min<size_t>(limit, 1024);
  me.graph_owner = ann_index_.get();
  me.graph_n = total_indexed_vector_;
  me.start_point = start_search_point_;
  me.off = offset_table_;
  me.nbr = neighbor_list_;
  me.T = num_threads_;
  me.L = L_master_;
  me.Lq = L_local_;
  me.I = subsearch_iterations_;
  me.prefilter = prefilter_enabled_;
  std::unique_lock<std::mutex> lk(dev.qmu);
  dev.queue.push_back(&me);
  while (!me.done) {
    if (dev.busy) {  // somebody is driving the device: wait until served, or until the device is free again
      dev.qcv.wait(lk, [&] { return me.done || !dev.busy; });
      continue;
    }
    dev.busy = true;  // become the leader: serve the head of the queue and everything compatible with it
    std::vector<Pending*> batch;
    Pending* head = dev.queue.front();
    for (auto it = dev.queue.begin(); it != dev.queue.end() && batch.size() < 256;) {
      if (SameKey(**it, *head)) {
        batch.push_back(*it);
        it = dev.queue.erase(it);
      } else {
        ++it;
      }
    }
    lk.unlock();
    try {
      RunBatch(dev, dimension_, batch);
    } catch (const std::exception& e) {   // (allocation failure while gathering the batch): the followers must still be released
      for (Pending* r : batch) r->error = std::string("gfx950 executor: ") + e.what();
    }
    lk.lock();
    for (Pending* r : batch) r->done = true;
    dev.busy = false;
    dev.qcv.notify_all();
  }
  lk.unlock();
  if (!me.error.empty()) throw std::runtime_error(me.error);
  if ((size_t)me.count > search_result_.size()) {
    search_result_.resize(me.count);
    distance_.resize(me.count);
  }
  for (int32_t i = 0; i < me.count; ++i) {
    search_result_[i] = me.ids[i];
    distance_[i] = me.dist[i];
  }
  result_size = me.count;
  return Status::OK();
}

// No vector arithmetic here: primary-key lookup / geo-index probe / full scan with the host filter engine.
Status VecSearchExecutor::SearchByAttribute(meta::TableSchema& table_schema, vectordb::engine::TableSegmentMVP* table_segment,
                                            const size_t skip, const size_t raw_limit, vectordb::Json& primary_keys,
                                            std::vector<ExprNodePtr>& filter_nodes, int64_t& result_size) {
  const int64_t total_vector = table_segment->record_number_;
  ConcurrentBitset& deleted = *(table_segment->deleted_);
  ExprEvaluator ev(filter_nodes, table_segment->field_name_mem_offset_map_, table_segment->primitive_offset_,
                   table_segment->var_len_attr_num_, table_segment->attribute_table_, table_segment->var_len_attr_table_);
  const int root = static_cast<int>(filter_nodes.size()) - 1;
  const int64_t limit = std::min<int64_t>((int64_t)raw_limit, total_vector);
  if (limit > (int64_t)search_result_.size()) search_result_.resize(limit);
  result_size = 0;
  int64_t seen = 0;  // rows that passed so far (before skip/limit windowing)
  // returns false once the window [skip, skip+limit) is full
  auto offer = [&](int64_t id) {
    if (deleted.test(id) || !ev.LogicalEvaluate(root, id)) return true;
    if (seen >= (int64_t)skip && seen < (int64_t)skip + limit) search_result_[result_size++] = id;
    ++seen;
    return seen < (int64_t)skip + limit;
  };
  const int64_t npk = primary_keys.GetSize();
  if (npk > 0) {
    for (int64_t i = 0; i < npk; ++i) {
      auto pk = primary_keys.GetArrayElement(i);
      size_t id = 0;
      if (table_segment->PK2ID(pk, id) && !offer((int64_t)id)) break;
    }
    return Status::OK();
  }
  bool used_geo = false;
  for (auto& field : table_schema.fields_) {
    if (field.field_type_ != meta::FieldType::GEO_POINT) continue;
    const int64_t node = ev.UpliftingGeoIndex(field.name_, root);
    if (node == -1) continue;
    used_geo = true;
    std::vector<vectordb::engine::index::GeospatialIndex::value_t> hits;
    const double lat = ev.NumEvaluate(filter_nodes[node]->arguments[1], -1, 0);
    const double lon = ev.NumEvaluate(filter_nodes[node]->arguments[2], -1, 0);
    const double rad = ev.NumEvaluate(filter_nodes[node]->arguments[3], -1, 0);
    table_segment->geospatial_indices_[field.name_]->searchWithinRadius(lat, lon, rad, hits);
    for (auto& h : hits)
      if (!offer(h.second)) break;
  }
  if (!used_geo)
    for (int64_t id = 0; id < total_vector; ++id)
      if (!offer(id)) break;
  return Status::OK();
}

}  // namespace execution
}  // namespace engine
}  // namespace vectordb
