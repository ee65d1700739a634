#!/bin/bash
# pieces-in-flight A/B (row_dists NL) of whole libraries: traversal 1M x 768 (prefilter on and off), build 1M x 768, headline flat line
R=$GRAFT_REPO_ROOT
cd $R
cp vectordb_amd/lib/libepsilla_gfx950.so /tmp/cur.so
for r in 1 2; do
for v in "$@"; do
  cp scripts/lab/_ab/$v.so vectordb_amd/lib/libepsilla_gfx950.so
  EPS_DEBUG=1 timeout 900 python scripts/bench_graph.py --rows 1000000 --dim 768 --data uniform --L 500 --T 1,4 --reps 5 --save-graph /tmp/g_nl.bin 2>/tmp/nl.err | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l)
    if 'kernel_ms' in j: print('$v', $r, 'on ', j['config'][-10:], 'kernel_ms %.3f' % j['kernel_ms'], 'qps %.0f' % j['qps'])"
  grep -E "Link \(|kNN graph" /tmp/nl.err | sed "s/^/$v $r /" | cut -c1-90
  EPS_TRV_PREFILTER=0 timeout 900 python scripts/bench_graph.py --rows 1000000 --dim 768 --data uniform --L 500 --T 4 --reps 5 --load-graph /tmp/g_nl.bin 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l)
    if 'kernel_ms' in j: print('$v', $r, 'off', j['config'][-10:], 'kernel_ms %.3f' % j['kernel_ms'], 'qps %.0f' % j['qps'])"
  timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --graph-rows 0 --configs none --recall-queries 128 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('$v', $r, 'flat ms/step %.3f' % j['ms_per_step'], 'kernel %.3f' % j['roofline']['kernel_ms_per_launch'], 'recall', j['recall_at_10'])"
done
done
cp /tmp/cur.so vectordb_amd/lib/libepsilla_gfx950.so
