#!/usr/bin/env python
"""Summarises a rocprofv3 (rocpd sqlite) --kernel-trace --stats database into a small CSV that can be
committed under profiles/:  python scripts/rocpd_summary.py <results.db> <out.csv> [top_n]"""
import csv
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    rows.sort(key=lambda r: -r[2])
    keep = [r for r in rows if r[0].startswith(("eps::", "_ZN3eps", "void eps::"))]
    others = [r for r in rows if r not in keep][:top]
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_ms", "pct_of_gpu_time", "group"])
        for r in keep:
            w.writerow([r[0][:160], r[1], "%.1f" % (r[2] / 1e3), "%.2f" % (r[3] / 1e3), "%.3f" % r[4], "libepsilla_gfx950"])
        for r in others:
            w.writerow([r[0][:160], r[1], "%.1f" % (r[2] / 1e3), "%.2f" % (r[3] / 1e3), "%.3f" % r[4], "harness (torch data gen / recall check)"])
    # per-dispatch detail of our kernels
    det = out.replace(".csv", "_dispatches.csv")
    cols = "name, start, duration, grid_x, grid_y, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, scratch_size"
    q = "select %s from kernels where name like '%%eps%%' order by start" % cols
    with open(det, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow([x.strip() for x in cols.split(",")])
        t0 = None
        for r in c.execute(q):
            t0 = r[1] if t0 is None else t0
            w.writerow([r[0][:80], (r[1] - t0) / 1e3, r[2] / 1e3] + list(r[3:]))


if __name__ == "__main__":
    main()
