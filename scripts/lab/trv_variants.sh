#!/bin/bash
# prefilter load-shape variants (libraries prebuilt under scripts/lab/_ab/trv_u<U8>_n<NL8>.so) on one graph: ROWS x 768 uniform
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
ROWS=${1:-1000000}
mkdir -p $O
cd $R
cp vectordb_amd/lib/libepsilla_gfx950.so /tmp/cur.so
EPS_TRV_PREFILTER=0 timeout 1500 python scripts/bench_graph.py --rows $ROWS --dim 768 --data uniform --L 500 --T 1,4 --reps 3 --save-graph /tmp/g_ab.bin 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l)
    if 'kernel_ms' in j: print('off', j['config'][-10:], 'kernel_ms %.3f' % j['kernel_ms'], 'qps %.0f' % j['qps'])
    elif 'seconds' in j: print(j)"
for v in scripts/lab/_ab/trv_u*.so; do
  cp $v vectordb_amd/lib/libepsilla_gfx950.so
  EPS_TRV_PREFILTER=1 timeout 600 python scripts/bench_graph.py --rows $ROWS --dim 768 --data uniform --L 500 --T 1,4 --reps 3 --load-graph /tmp/g_ab.bin 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l)
    if 'kernel_ms' in j: print('$v'[-12:], j['config'][-10:], 'kernel_ms %.3f' % j['kernel_ms'], 'qps %.0f' % j['qps'], 'fp32 rows %.0f' % j['fp32_rows_per_query'], 'recall %.4f' % j['recall_at_10'])"
done
cp /tmp/cur.so vectordb_amd/lib/libepsilla_gfx950.so
