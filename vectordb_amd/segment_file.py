"""Writes a table in the reference's on-disk form: `<db_path>/catalog` + `<db_path>/<table_id>/data_mvp.bin`,
byte-compatible with TableSegmentMVP::SaveTableSegment (reference: engine/db/table_segment_mvp.cpp:939-1010) and the catalog
DBServer::LoadDB reads (engine/db/catalog/basic_meta_impl.cpp), so that the UNCHANGED loader of the reference (its
TableSegmentMVP file constructor, table_segment_mvp.cpp:133-281, which also rebuilds the primary-key index) brings up millions
of rows in seconds - the reference's only ingest path is JSON through insert() plus a text WAL of every float
(table_mvp.cpp:272-276), hours at 10M x 768.

Primitive fields (INT1/2/4/8, FLOAT, DOUBLE, BOOL) and dense VECTOR_FLOAT fields; no strings / JSON / sparse vectors.
Pure host code (numpy); nothing here touches the device."""
import json
import os

import numpy as np

# meta::FieldType / MetricType values (engine/db/catalog/meta_types.hpp:20-52)
FIELD_TYPES = {"TINYINT": (1, "<i1"), "SMALLINT": (2, "<i2"), "INT": (3, "<i4"), "BIGINT": (4, "<i8"), "FLOAT": (10, "<f4"),
               "DOUBLE": (11, "<f8"), "BOOL": (30, "?"), "VECTOR_FLOAT": (40, None)}
METRICS = {"EUCLIDEAN": 1, "COSINE": 2, "DOT_PRODUCT": 3}


def write_database(db_path, table_name, fields, columns, table_id=0, db_id=0, deleted=None, chunk_rows=1 << 18):
    """fields: list of dicts as create_table takes them ({"name", "dataType", "primaryKey"?, "dimensions"?, "metricType"?});
    columns: {name: array} - one 1-D array per primitive field, one [n][dim] float32 array (numpy, or anything with
    .shape and slicing that np.asarray accepts per slice, e.g. a CPU torch tensor) per vector field.  COSINE fields must
    already be normalised (the reference normalises at insert, table_segment_mvp.cpp:574-587).  Returns the row count."""
    n = None
    prims, vecs, cat_fields = [], [], []
    for fid, f in enumerate(fields):
        code, dt = FIELD_TYPES[f["dataType"]]
        col = columns[f["name"]]
        rows = int(col.shape[0])
        n = rows if n is None else n
        if rows != n:
            raise ValueError("column %s has %d rows, expected %d" % (f["name"], rows, n))
        cf = {"field_type": code, "id": fid, "is_index_field": False, "is_primary_key": bool(f.get("primaryKey", False)), "name": f["name"]}
        if dt is None:
            if int(col.shape[1]) != int(f["dimensions"]):
                raise ValueError("column %s: dimension mismatch" % f["name"])
            cf["metric_type"] = METRICS[f.get("metricType", "EUCLIDEAN")]
            cf["vector_dimension"] = int(f["dimensions"])
            vecs.append(col)
        else:
            prims.append((np.dtype(dt), col))
        cat_fields.append(cf)
    os.makedirs(os.path.join(db_path, str(table_id)), exist_ok=True)
    with open(os.path.join(db_path, "catalog"), "w") as fh:
        json.dump({"id": db_id, "tables": [{"fields": cat_fields, "id": table_id, "name": table_name}]}, fh, separators=(",", ":"))
    row_dt = np.dtype([("f%d" % i, dt) for i, (dt, _) in enumerate(prims)]) if prims else None   # packed, schema order (Init, :51-92)
    bits = np.zeros((n + 7) // 8, np.uint8) if deleted is None else np.ascontiguousarray(deleted, np.uint8)
    if bits.size < (n + 7) // 8:
        raise ValueError("deleted bitset shorter than ceil(n/8) bytes")
    path = os.path.join(db_path, str(table_id), "data_mvp.bin")
    with open(path + ".tmp", "wb") as fh:
        fh.write(np.array([n, 0, bits.size], np.int64).tobytes())      # record_number, first_record_id, bitset_size
        fh.write(bits.tobytes())
        if row_dt is not None:
            for s in range(0, n, chunk_rows):
                e = min(n, s + chunk_rows)
                rows = np.empty(e - s, row_dt)
                for i, (dt, col) in enumerate(prims):
                    rows["f%d" % i] = np.asarray(col[s:e]).astype(dt, copy=False)
                fh.write(rows.tobytes())
        for col in vecs:                                                   # (no variable-length attributes in between)
            for s in range(0, n, chunk_rows):
                fh.write(np.ascontiguousarray(np.asarray(col[s:min(n, s + chunk_rows)]), dtype="<f4").tobytes())
        fh.write(np.array([-1], np.int64).tobytes())                       # wal_global_id_: nothing consumed yet
    os.replace(path + ".tmp", path)
    return n
