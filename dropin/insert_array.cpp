// epsdrop::InsertArray (include/epsdrop/insert_array.hpp): the column-buffer image of DBServer::Insert -> TableMVP::Insert ->
// TableSegmentMVP::Insert (engine/db/db_server.cpp:266-280, db/table_mvp.cpp:272-276, db/table_segment_mvp.cpp:455-808).
#include "epsdrop/insert_array.hpp"

#include <cmath>
#include <cstring>
#include <algorithm>
#include <mutex>
#include <thread>

#include "db/table_mvp.hpp"
#include "db/table_segment_mvp.hpp"
#include "utils/error.hpp"

namespace epsdrop {

using vectordb::Status;
using vectordb::engine::TableSegmentMVP;
using vectordb::engine::UniqueKey;
namespace meta = vectordb::engine::meta;

// ---- the segment's two private members the insert path needs (its update mutex and its primary-key map), reached without touching the
// reference's header: an explicit instantiation may name private members ([temp.spec]/6), and hands the pointers-to-member out through a friend
namespace {
template <typename Tag, typename Tag::type M>
struct Expose {
  friend typename Tag::type get(Tag) { return M; }
};
struct SegMutex {
  typedef std::mutex TableSegmentMVP::*type;
  friend type get(SegMutex);
};
struct SegKeys {
  typedef UniqueKey TableSegmentMVP::*type;
  friend type get(SegKeys);
};
template struct Expose<SegMutex, &TableSegmentMVP::data_update_mutex_>;
template struct Expose<SegKeys, &TableSegmentMVP::primary_key_>;

inline int64_t as_int(const ColumnView& c, int64_t i) {
  switch (c.kind) {
    case ColumnView::I8: return static_cast<const int8_t*>(c.data)[i];
    case ColumnView::I16: return static_cast<const int16_t*>(c.data)[i];
    case ColumnView::I32: return static_cast<const int32_t*>(c.data)[i];
    case ColumnView::I64: return static_cast<const int64_t*>(c.data)[i];
    case ColumnView::U8: return static_cast<const uint8_t*>(c.data)[i];
    case ColumnView::F32: return (int64_t)static_cast<const float*>(c.data)[i];
    case ColumnView::F64: return (int64_t)static_cast<const double*>(c.data)[i];
    default: return 0;
  }
}
inline double as_double(const ColumnView& c, int64_t i) {
  switch (c.kind) {
    case ColumnView::F32: return (double)static_cast<const float*>(c.data)[i];
    case ColumnView::F64: return static_cast<const double*>(c.data)[i];
    default: return (double)as_int(c, i);
  }
}
// rows [0, n) in contiguous chunks on a few short-lived threads when there is enough to copy (an OpenMP region here measured 0.4-1 s per call
// while another runtime's workers were spinning in the process: the copy is memory-bound, a handful of plain threads is all it needs)
template <typename F>
void for_rows(int64_t n, int64_t bytes_per_row, F&& body) {
  const int64_t total = n * std::max<int64_t>(bytes_per_row, 1);
  int64_t nt = std::min<int64_t>(total >> 22, std::min<int64_t>(16, (int64_t)std::thread::hardware_concurrency() / 2));   // one thread per 4 MB
  if (nt <= 1) {
    body((int64_t)0, n);
    return;
  }
  std::vector<std::thread> pool;
  const int64_t chunk = (n + nt - 1) / nt;
  for (int64_t t = 1; t < nt; ++t) pool.emplace_back([&, t] { body(std::min(n, t * chunk), std::min(n, (t + 1) * chunk)); });
  body((int64_t)0, std::min(n, chunk));
  for (auto& th : pool) th.join();
}
inline bool is_int_kind(ColumnView::Kind k) { return k == ColumnView::I8 || k == ColumnView::I16 || k == ColumnView::I32 || k == ColumnView::I64 || k == ColumnView::U8; }
inline bool is_float_kind(ColumnView::Kind k) { return k == ColumnView::F32 || k == ColumnView::F64; }
}  // namespace

Status InsertArray(vectordb::engine::DBServer& server, const std::string& db_name, const std::string& table_name, const std::vector<ColumnView>& columns,
                   int64_t n, bool upsert, bool sync, InsertArrayResult* result) {
  if (result) *result = InsertArrayResult();
  // ---- DBServer::Insert (db_server.cpp:271-279)
  auto db = server.GetDB(db_name);
  if (db == nullptr) return Status(vectordb::DB_UNEXPECTED_ERROR, "DB not found: " + db_name);
  auto table = db->GetTable(table_name);
  if (table == nullptr) return Status(vectordb::DB_UNEXPECTED_ERROR, "Table not found: " + table_name);
  if (n < 0) return Status(vectordb::INVALID_PAYLOAD, "InsertArray: negative record count");
  TableSegmentMVP& seg = *table->table_segment_;
  meta::TableSchema& schema = table->table_schema_;
  if (!schema.indices_.empty()) return Status(vectordb::INVALID_PAYLOAD, "InsertArray: the table has embedding indices (fields filled by the embedding service); use insert");

  // ---- every schema field has a column of a matching shape (the JSON path: "Record i missing field", table_segment_mvp.cpp:466-474)
  std::vector<const ColumnView*> col_of(schema.fields_.size(), nullptr);
  int pk_field = -1;
  for (size_t f = 0; f < schema.fields_.size(); ++f) {
    auto& field = schema.fields_[f];
    if (field.is_index_field_) continue;
    for (auto& c : columns)
      if (c.name == field.name_) col_of[f] = &c;
    if (!col_of[f]) return Status(vectordb::INVALID_RECORD, "Record 0 missing field: " + field.name_);
    const ColumnView& c = *col_of[f];
    switch (field.field_type_) {
      case meta::FieldType::INT1: case meta::FieldType::INT2: case meta::FieldType::INT4: case meta::FieldType::INT8: case meta::FieldType::BOOL:
        if (!is_int_kind(c.kind) || c.width != 1 || (n > 0 && !c.data)) return Status(vectordb::INVALID_PAYLOAD, "InsertArray: field " + field.name_ + " wants a 1-D integer column");
        break;
      case meta::FieldType::FLOAT: case meta::FieldType::DOUBLE:
        if (c.kind == ColumnView::STR || c.width != 1 || (n > 0 && !c.data)) return Status(vectordb::INVALID_PAYLOAD, "InsertArray: field " + field.name_ + " wants a 1-D numeric column");
        break;
      case meta::FieldType::STRING:
        if (c.kind != ColumnView::STR || !c.strings || (int64_t)c.strings->size() != n) return Status(vectordb::INVALID_PAYLOAD, "InsertArray: field " + field.name_ + " wants a list of n strings");
        break;
      case meta::FieldType::VECTOR_FLOAT: case meta::FieldType::VECTOR_DOUBLE:
        // (the JSON path skips a record whose vector has the wrong length, :566-571; a column has ONE width: refused as a whole)
        if (!is_float_kind(c.kind) || (n > 0 && !c.data) || c.width != (int64_t)field.vector_dimension_)
          return Status(vectordb::INVALID_PAYLOAD, "InsertArray: field " + field.name_ + " wants a 2-D float32 / float64 column of width " + std::to_string(field.vector_dimension_));
        break;
      default:
        return Status(vectordb::INVALID_PAYLOAD, "InsertArray: field " + field.name_ + " (JSON, GEO_POINT or sparse vector): use insert");
    }
    if (field.is_primary_key_) pk_field = (int)f;
  }

  std::unique_lock<std::mutex> lock(seg.*get(SegMutex()));
  UniqueKey& keys = seg.*get(SegKeys());
  if (n == 0) return Status::OK();
  // ---- capacity (:476-482)
  if (seg.record_number_ + (size_t)n > seg.size_limit_)
    return Status(vectordb::DB_UNEXPECTED_ERROR, "Currently, each table in this database can hold up to " + std::to_string(seg.size_limit_) + " records. " +
                                                     "To insert more records, please unload the database and reload with a larger vectorScale parameter.");

  // ---- pass 1, in record order (as the JSON loop meets them): the primary keys decide every record's slot.  A duplicate key skips its record
  // WITHOUT advancing the cursor; upsert takes the slot and remembers the row it replaces (:653-736)
  const size_t first = seg.record_number_;
  size_t cursor = first;
  std::vector<int64_t> slot((size_t)n);
  size_t skipped = 0, upsert_size = 0;
  std::vector<int64_t> upd_int;
  std::vector<std::string> upd_str;
  std::vector<size_t> upd_old, upd_new;
  const meta::FieldType pk_type = pk_field >= 0 ? schema.fields_[(size_t)pk_field].field_type_ : meta::FieldType::UNKNOWN;
  if (pk_field >= 0) {
    const ColumnView& c = *col_of[(size_t)pk_field];
    auto one = [&](auto key, int64_t i) {
      const bool exist = !keys.addKeyIfNotExist(key, cursor);
      if (exist) {
        if (!upsert) {
          slot[(size_t)i] = -1;
          ++skipped;
          return;
        }
        size_t old_idx = 0;
        keys.getKey(key, old_idx);
        if constexpr (std::is_same<decltype(key), std::string>::value) upd_str.push_back(key); else upd_int.push_back((int64_t)key);
        upd_old.push_back(old_idx);
        upd_new.push_back(cursor);
        ++upsert_size;
      }
      slot[(size_t)i] = (int64_t)cursor++;
    };
    for (int64_t i = 0; i < n; ++i) {
      switch (pk_type) {
        case meta::FieldType::STRING: one((*c.strings)[(size_t)i], i); break;
        case meta::FieldType::INT1: one(static_cast<int8_t>(as_int(c, i)), i); break;
        case meta::FieldType::INT2: one(static_cast<int16_t>(as_int(c, i)), i); break;
        case meta::FieldType::INT4: one(static_cast<int32_t>(as_int(c, i)), i); break;
        case meta::FieldType::INT8: one(static_cast<int64_t>(as_int(c, i)), i); break;
        default: slot[(size_t)i] = (int64_t)cursor++;   // (a key of another type is not registered, :653-736)
      }
    }
  } else {
    for (int64_t i = 0; i < n; ++i) slot[(size_t)i] = (int64_t)cursor++;
  }

  // ---- pass 2: the accepted records' fields, every record to its own slot (any order: the slots are disjoint)
  for (size_t f = 0; f < schema.fields_.size(); ++f) {
    auto& field = schema.fields_[f];
    if (field.is_index_field_) continue;
    const ColumnView& c = *col_of[f];
    const size_t off = seg.field_id_mem_offset_map_[field.id_];
    if (field.field_type_ == meta::FieldType::STRING) {
      auto& column = seg.var_len_attr_table_[off];
      for (int64_t i = 0; i < n; ++i)
        if (slot[(size_t)i] >= 0) column[(size_t)slot[(size_t)i]] = (*c.strings)[(size_t)i];
    } else if (field.field_type_ == meta::FieldType::VECTOR_FLOAT || field.field_type_ == meta::FieldType::VECTOR_DOUBLE) {
      float* tab = seg.vector_tables_[off];
      const int64_t dim = seg.vector_dims_[off];
      const bool cosine = field.metric_type_ == meta::MetricType::COSINE;
      for_rows(n, dim * 4, [&](int64_t lo, int64_t hi) {
      for (int64_t i = lo; i < hi; ++i) {
        if (slot[(size_t)i] < 0) continue;
        float* dst = tab + (size_t)slot[(size_t)i] * dim;
        float sum = 0;
        if (c.kind == ColumnView::F32) {
          const float* src = static_cast<const float*>(c.data) + (size_t)i * dim;
          for (int64_t j = 0; j < dim; ++j) {
            float value = src[j];
            sum += value * value;
            dst[j] = value;
          }
        } else {
          const double* src = static_cast<const double*>(c.data) + (size_t)i * dim;
          for (int64_t j = 0; j < dim; ++j) {
            float value = static_cast<float>(src[j]);
            sum += value * value;
            dst[j] = value;
          }
        }
        if (cosine && sum > 1e-10) {   // (:574-587)
          sum = std::sqrt(sum);
          for (int64_t j = 0; j < dim; ++j) dst[j] /= sum;
        }
      }
      });
    } else {
      char* base = seg.attribute_table_ + off;
      const int64_t stride = seg.primitive_offset_;
      for (int64_t i = 0; i < n; ++i) {
        if (slot[(size_t)i] < 0) continue;
        char* at = base + (size_t)slot[(size_t)i] * stride;
        switch (field.field_type_) {
          case meta::FieldType::INT1: { int8_t v = static_cast<int8_t>(as_int(c, i)); std::memcpy(at, &v, sizeof v); break; }
          case meta::FieldType::INT2: { int16_t v = static_cast<int16_t>(as_int(c, i)); std::memcpy(at, &v, sizeof v); break; }
          case meta::FieldType::INT4: { int32_t v = static_cast<int32_t>(as_int(c, i)); std::memcpy(at, &v, sizeof v); break; }
          case meta::FieldType::INT8: { int64_t v = as_int(c, i); std::memcpy(at, &v, sizeof v); break; }
          case meta::FieldType::FLOAT: { float v = static_cast<float>(as_double(c, i)); std::memcpy(at, &v, sizeof v); break; }
          case meta::FieldType::DOUBLE: { double v = as_double(c, i); std::memcpy(at, &v, sizeof v); break; }
          case meta::FieldType::BOOL: { bool v = as_int(c, i) != 0; std::memcpy(at, &v, sizeof v); break; }
          default: break;
        }
      }
    }
  }

  // ---- publish (:759-798)
  seg.record_number_.store(cursor);
  if (upsert) {
    for (size_t idx = 0; idx < upsert_size; ++idx) {
      switch (pk_type) {
        case meta::FieldType::INT1: keys.updateKey(static_cast<int8_t>(upd_int[idx]), upd_new[idx]); break;
        case meta::FieldType::INT2: keys.updateKey(static_cast<int16_t>(upd_int[idx]), upd_new[idx]); break;
        case meta::FieldType::INT4: keys.updateKey(static_cast<int32_t>(upd_int[idx]), upd_new[idx]); break;
        case meta::FieldType::INT8: keys.updateKey(static_cast<int64_t>(upd_int[idx]), upd_new[idx]); break;
        case meta::FieldType::STRING: keys.updateKey(upd_str[idx], upd_new[idx]); break;
        default: break;
      }
      seg.deleted_->set(upd_old[idx]);
    }
  }
  seg.skip_sync_disk_.store(false);
  lock.unlock();
  if (result) {
    result->inserted = n - (int64_t)skipped;
    result->skipped = (int64_t)skipped;
  }
  // ---- no write-ahead-log record: durable from the next segment flush on; sync = that flush now (what TableMVP::Dump does first, table_mvp.cpp:607)
  if (sync) return seg.SaveTableSegment(schema, table->db_catalog_path_, true);
  return Status::OK();
}

}  // namespace epsdrop
