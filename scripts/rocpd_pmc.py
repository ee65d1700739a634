#!/usr/bin/env python
"""Per-dispatch PMC table from rocprofv3 --pmc databases: python scripts/rocpd_pmc.py out.csv db1 [db2 ...]"""
import csv
import sqlite3
import sys
from collections import OrderedDict

out, dbs = sys.argv[1], sys.argv[2:]
rows = OrderedDict()
counters = []
for db in dbs:
    c = sqlite3.connect(db)
    q = "select dispatch_id, kernel_name, grid_size, counter_name, sum(value), max(duration) from counters_collection group by dispatch_id, counter_name order by dispatch_id"
    order = {}
    for did, kn, grid, cn, val, dur in c.execute(q):
        # dispatch ids differ between runs: key on (kernel, ordinal of that kernel within its run)
        key_base = kn.split("(")[0][-48:]
        o = order.setdefault((db, did), None)
        rows.setdefault((db, did, key_base, grid), {})[cn] = (val, dur)
        if cn not in counters:
            counters.append(cn)
# align runs by (kernel name, occurrence index)
merged = OrderedDict()
for db in dbs:
    occ = {}
    for (d, did, kn, grid), vals in rows.items():
        if d != db:
            continue
        i = occ.get(kn, 0)
        occ[kn] = i + 1
        m = merged.setdefault((kn, i, grid), {})
        for cn, (val, dur) in vals.items():
            m[cn] = val
            m.setdefault("duration_us", dur / 1e3)
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "occurrence", "grid_threads", "duration_us"] + counters)
    for (kn, i, grid), m in merged.items():
        w.writerow([kn, i, grid, "%.1f" % m.get("duration_us", 0)] + ["%.0f" % m.get(cn, float("nan")) for cn in counters])
print(open(out).read())
