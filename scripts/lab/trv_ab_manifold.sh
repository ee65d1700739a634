#!/bin/bash
# traversal A/B of whole libraries on low-intrinsic-dimension rows (ROWS x 768 manifold): short searches, where fixed per-query
# costs show.   usage: trv_ab_manifold.sh ROWS name1 name2 ...   (scripts/lab/_ab/<name>.so)
R=$GRAFT_REPO_ROOT
cd $R
ROWS=$1; shift
cp vectordb_amd/lib/libepsilla_gfx950.so /tmp/cur.so
for r in 1 2; do
for v in "$@"; do
  cp scripts/lab/_ab/$v.so vectordb_amd/lib/libepsilla_gfx950.so
  timeout 900 python scripts/bench_graph.py --rows $ROWS --dim 768 --data manifold --L 50,100 --T 1,4 --reps 5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l)
    if 'kernel_ms' in j: print('$v', $r, j['config'][-10:], 'kernel_ms %.3f' % j['kernel_ms'], 'qps %.0f' % j['qps'], 'recall %.4f' % j['recall_at_10'], 'fp32', j.get('fp32_rows_per_query'))"
done
done
cp /tmp/cur.so vectordb_amd/lib/libepsilla_gfx950.so
