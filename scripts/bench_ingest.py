#!/usr/bin/env python
"""Measurements for the SURVEY 8f "next" rows on the data side of the path (JSON lines):

  f2  append fast path: a table with a warm fp16 mirror receives `--append` more rows (host memory, as the DBMS hands them
      over); time of eps_index_append_rows, of the first batched search after it (the mirror is EXTENDED by the new rows
      only) and, beside it, of the same search on a freshly attached index (mirror rebuilt from scratch).
  f3  eps_index_load_table: a `data_mvp.bin` in the reference's layout (table_segment_mvp.cpp:939-1010: record count,
      first record id, deleted bitset, fixed-width attribute rows, variable-length attributes, dense vector fields, WAL id),
      written here by numpy (the layout is pinned against a file the reference wrote in tests/test_gpu_parity.py), read straight
      into HBM.

    python scripts/bench_ingest.py [--rows 10000000] [--dim 768] [--append 100000] [--load-rows 1000000] [--batch 1024]"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("EPS_TUNING_FROM_ENV", "1")   # (scripts steer the library's engine switches through the environment: vectordb_amd/_lib.py)
import vectordb_amd as amd  # noqa: E402


def gen(n, d, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    X = torch.empty((n, d), device="cuda")
    for s in range(0, n, 1 << 20):
        e = min(n, s + (1 << 20))
        X[s:e] = torch.rand((e - s, d), generator=g, device="cuda")
    return X


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--append", type=int, default=100_000)
    ap.add_argument("--load-rows", type=int, default=1_000_000)
    ap.add_argument("--batch", type=int, default=1024)
    args = ap.parse_args()
    n, d, m, b = args.rows, args.dim, args.append, args.batch

    # ---- f2
    Xh = gen(n + m, d, 42).cpu().numpy()     # the table lives in host memory, as the DBMS holds it: the index owns its device copy
    torch.cuda.empty_cache()
    Q = gen(b, d, 43).cpu().numpy()          # host queries / host results (3 MB + 120 KB over PCIe are inside the search times below)
    tail_host = Xh[n:]                       # the appended rows arrive in host memory
    ix = amd.GpuIndex(d, 0).use_torch_stream()
    att_ms, _ = timed(lambda: ix.attach_rows(Xh[:n]))
    kw = dict(mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA)
    first_ms, _ = timed(lambda: ix.search(Q, 10, **kw))          # builds the fp16 mirror of n rows
    warm_ms, _ = timed(lambda: ix.search(Q, 10, **kw))
    app_ms, _ = timed(lambda: ix.append_rows(tail_host))
    after_ms, got = timed(lambda: ix.search(Q, 10, **kw))        # extends the mirror by m rows, then searches n + m
    warm2_ms, _ = timed(lambda: ix.search(Q, 10, **kw))
    ref = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
    same = bool(np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]))
    ix.close()
    print(json.dumps({"config": "f2 append %d rows to %d x %d (warm fp16 mirror), batch %d" % (m, n, d, b),
                      "attach_rows_host_to_hbm_ms": att_ms, "attach_GBps": n * d * 4 / (att_ms * 1e-3) / 1e9, "first_search_builds_mirror_ms": first_ms, "warm_search_ms": warm_ms, "append_rows_ms": app_ms,
                      "append_GBps_host_to_hbm": m * d * 4 / (app_ms * 1e-3) / 1e9,
                      "first_search_after_append_ms": after_ms, "mirror_extension_cost_ms": after_ms - warm2_ms,
                      "full_mirror_build_cost_ms": first_ms - warm_ms, "warm_search_after_ms": warm2_ms,
                      "equals_exact_stream_scan": same}), flush=True)
    del Xh
    torch.cuda.empty_cache()

    # ---- f3
    ln, dims = args.load_rows, [d]
    Xl = gen(ln, d, 44).cpu().numpy()
    rng = np.random.default_rng(45)
    ids = np.arange(ln, dtype=np.int64)
    price = rng.random(ln).astype(np.float32)
    attrs = np.zeros(ln, dtype=np.dtype([("id", "<i8"), ("price", "<f4")]))     # primitive_offset = 12
    attrs["id"], attrs["price"] = ids, price
    bits = np.zeros((ln + 7) // 8, dtype=np.uint8)
    for r in (3, 77, ln - 1):
        bits[r >> 3] |= np.uint8(1 << (r & 7))
    path = os.path.join(tempfile.mkdtemp(prefix="eps_ingest_"), "data_mvp.bin")
    t0 = time.perf_counter()
    with open(path, "wb") as f:
        f.write(np.array([ln], dtype="<u8").tobytes())
        f.write(np.array([0, bits.size], dtype="<i8").tobytes())
        f.write(bits.tobytes())
        f.write(attrs.tobytes())
        f.write(Xl.tobytes())
        f.write(np.array([0], dtype="<i8").tobytes())
    write_s = time.perf_counter() - t0
    fsize = os.path.getsize(path)
    ixl = amd.GpuIndex(d, 0).use_torch_stream()
    load_ms, rows = timed(lambda: ixl.load_table(path, primitive_offset=12, var_len_attrs=0, dense_dims=dims, field=0))
    load2_ms, _ = timed(lambda: ixl.load_table(path, primitive_offset=12, var_len_attrs=0, dense_dims=dims, field=0))
    ixl.set_filter_program([("f32", 8), ("const", 0.5), ("<",)])          # Price < 0.5 over the attribute rows the loader kept
    Ql = Xl[[3, 10, 77, 500]]                                             # rows 3 and 77 are deleted, row 10 / 500 pass or fail the filter
    idsr, dist, cnt = ixl.search(Ql, 1, mode=amd.MODE_FLAT)
    ok = bool(idsr[0][0] != 3 and idsr[2][0] != 77 and (idsr[1][0] == 10) == bool(price[10] < 0.5) and (idsr[3][0] == 500) == bool(price[500] < 0.5))
    ixl.close()
    os.remove(path)
    print(json.dumps({"config": "f3 eps_index_load_table %d x %d data_mvp.bin (%.2f GB, page cache warm)" % (ln, d, fsize / 1e9),
                      "rows": int(rows), "load_ms": load_ms, "load_again_ms": load2_ms, "GBps_file_to_hbm": fsize / (load2_ms * 1e-3) / 1e9,
                      "file_written_in_s": write_s, "deleted_and_filter_respected": ok}), flush=True)


if __name__ == "__main__":
    main()
