#!/bin/bash
# A/B of whole libraries on one box: bench.py (headline config) with every scripts/lab/_ab/<name>.so given, alternating.
#   usage: ab_libs.sh rounds name1 name2 ...
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out/ab
cp vectordb_amd/lib/libepsilla_gfx950.so /tmp/cur.so
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    cp scripts/lab/_ab/$v.so vectordb_amd/lib/libepsilla_gfx950.so
    timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --graph-rows 0 --configs none --recall-queries 128 --no-pmc 2> gpurun_out/ab/$v.$r.err | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('$v', $r, 'ms/step %.3f' % j['ms_per_step'], 'kernel %.3f' % j['roofline']['kernel_ms_per_launch'], 'frac %.4f' % j['roofline']['frac'], 'recall', j['recall_at_10'], 'rerank', j['stats']['rerank_rows_per_query'], 'ovf', j['stats']['overflow_queries'])"
  done
done
cp /tmp/cur.so vectordb_amd/lib/libepsilla_gfx950.so
