#!/bin/bash
R=$GRAFT_REPO_ROOT
ROWS=${1:-1000000}
cd $R
timeout 1500 python scripts/bench_graph.py --rows $ROWS --dim 768 --data uniform --L 500 --T 4 --reps 1 --save-graph /tmp/g_ll.bin > /dev/null 2>&1
for L in 2000 4000 8000; do
for kn in "X=0" "EPS_TRV_LDS_KB=64 EPS_TRV_WAVES=4" "EPS_TRV_LDS_KB=64 EPS_TRV_WAVES=8" "EPS_TRV_LDS_KB=64 EPS_TRV_WAVES=16" "EPS_TRV_LDS_KB=64 EPS_TRV_WAVES=8 EPS_TRV_PER_CU=3"; do
  env $kn timeout 600 python scripts/bench_graph.py --rows $ROWS --dim 768 --data uniform --L $L --T 4 --reps 2 --load-graph /tmp/g_ll.bin 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.readline()); print('L=$L', '$kn'.ljust(60), 'kernel_ms %.2f' % j['kernel_ms'], 'qps %.0f' % j['qps'])"
done
done
