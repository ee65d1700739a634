#!/bin/bash
# A/B of the flat-scan filter kernel on one box: bench.py with the library as built, then with scripts/lab/_ab/libepsilla_gfx950_base.so
# swapped in (the box works on a copy of the tree), alternating so that clock drift shows.   usage: ab_flat.sh [rounds]
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out/ab
cp vectordb_amd/lib/libepsilla_gfx950.so /tmp/new.so
cp scripts/lab/_ab/libepsilla_gfx950_base.so /tmp/base.so
for r in $(seq 1 ${1:-2}); do
  for v in new base; do
    cp /tmp/$v.so vectordb_amd/lib/libepsilla_gfx950.so
    timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --graph-rows 0 --recall-queries 128 2> gpurun_out/ab/$v.$r.err | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('$v', $r, 'ms/step %.3f' % j['ms_per_step'], 'kernel %.3f' % j['roofline']['kernel_ms_per_launch'], 'frac %.4f' % j['roofline']['frac'], 'recall', j['recall_at_10'], 'rerank', j['stats']['rerank_rows_per_query'], 'ovf', j['stats']['overflow_queries'])"
  done
done
cp /tmp/new.so vectordb_amd/lib/libepsilla_gfx950.so
