#!/bin/bash
# traversal launch-shape knobs with the prefilter on (ROWS x 768 uniform, T=4 L=500)
R=$GRAFT_REPO_ROOT
ROWS=${1:-1000000}
cd $R
EPS_TRV_PREFILTER=1 timeout 1500 python scripts/bench_graph.py --rows $ROWS --dim 768 --data uniform --L 500 --T 4 --reps 3 --save-graph /tmp/g_ab.bin 2>/dev/null | tail -1 | cut -c1-400
for kn in "EPS_TRV_WAVES=8" "EPS_TRV_WAVES=16" "EPS_TRV_PER_CU=3" "EPS_TRV_PER_CU=5" "EPS_TRV_PER_CU=6" "EPS_TRV_PER_CU=8"; do
  echo $kn
  env $kn EPS_TRV_PREFILTER=1 timeout 600 python scripts/bench_graph.py --rows $ROWS --dim 768 --data uniform --L 500 --T 4 --reps 3 --load-graph /tmp/g_ab.bin 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.readline()); print('kernel_ms %.3f' % j['kernel_ms'], 'qps %.0f' % j['qps'])"
done
EPS_TRV_PROF=1 EPS_TRV_PREFILTER=1 timeout 600 python scripts/bench_graph.py --rows $ROWS --dim 768 --data uniform --L 500 --T 4 --reps 1 --load-graph /tmp/g_ab.bin 2>&1 | grep "eps trv" | tail -13
