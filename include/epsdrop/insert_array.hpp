// C++-level bulk ingest of the drop-in ("binary / NumPy bulk insert bypassing per-float JSON").
//
// The reference's only ingest is DBServer::Insert (engine/db/db_server.cpp:266-280) -> TableMVP::Insert (db/table_mvp.cpp:272-276: the
// whole batch is dumped to TEXT and written to the write-ahead log, db/wal/write_ahead_log.hpp:71-91) -> TableSegmentMVP::Insert
// (db/table_segment_mvp.cpp:455-808: every float of every vector is fetched from a JSON array element).  At 10M x 768 that is hours
// and ~100 GB of log.  InsertArray writes the same segment state from COLUMN buffers - what TableSegmentMVP::Insert would have left
// behind for the same records, bit for bit:
//   * capacity check first, same Status and text (:476-482);
//   * per record, fields in schema order at `cursor`; dense vectors cast to float, |v|^2 summed in float in index order, COSINE rows
//     with |v|^2 > 1e-10 divided by sqrt(|v|^2) element by element (:564-587); primitive fields cast as the JSON path casts them (:589-627);
//   * the primary key is registered AFTER the record's fields were written; a duplicate key skips the record without advancing the
//     cursor (the next record overwrites its slot), or - upsert - replaces the key's row and marks the old one deleted (:653-792);
//   * record_number_ published once at the end; skip_sync_disk_ cleared; "inserted" / "skipped" counted as the reference reports them.
// What it does NOT do: no write-ahead-log record (the rows are durable from the next segment flush on - DBServer's periodic flush,
// unload, Rebuild, or `sync = true` here, which calls TableMVP::Dump's segment save before returning); tables with JSON, GEO_POINT or
// sparse-vector fields, or with embedding indices (fields the embedding service fills), are refused with INVALID_PAYLOAD - they
// keep the reference's Insert.  The device mirror of the table picks the new rows up on the next search (tail upload,
// dropin/vec_search_executor.cpp) - nothing of the existing rows crosses PCIe again.
// Additive: nothing of the reference's classes changes (the segment's private members are reached through an explicit template
// instantiation, dropin/insert_array.cpp).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "db/db_server.hpp"
#include "utils/status.hpp"

namespace epsdrop {

struct ColumnView {
  enum Kind { I8, I16, I32, I64, U8, F32, F64, STR };
  std::string name;                               // schema field name
  Kind kind = F32;
  const void* data = nullptr;                     // numeric kinds: row-major [n][width] of the kind's C type
  int64_t width = 1;                              // 1 for scalar fields, the vector dimension for dense vector fields
  const std::vector<std::string>* strings = nullptr;   // STR: n strings
};

struct InsertArrayResult {
  int64_t inserted = 0, skipped = 0;
};

vectordb::Status InsertArray(vectordb::engine::DBServer& server, const std::string& db_name, const std::string& table_name,
                             const std::vector<ColumnView>& columns, int64_t n, bool upsert, bool sync, InsertArrayResult* result);

}  // namespace epsdrop
