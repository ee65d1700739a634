#!/bin/bash
# T = 4, L = 2000 (queues of ~96 KB): one workgroup of 16 wavefronts per CU with the queues in LDS (default) against queues in HBM with
# four workgroups per CU (EPS_TRV_LDS_KB=64), ROWS x 768 uniform
R=$GRAFT_REPO_ROOT
ROWS=${1:-1000000}
cd $R
timeout 1500 python scripts/bench_graph.py --rows $ROWS --dim 768 --data uniform --L 2000 --T 1,4 --reps 3 --save-graph /tmp/g_ll.bin 2>/dev/null | tail -2 | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('default      ', j['config'][-11:], 'kernel_ms %.2f' % j['kernel_ms'], 'qps %.0f' % j['qps'])"
for kn in "EPS_TRV_LDS_KB=64" "EPS_TRV_LDS_KB=64 EPS_TRV_WAVES=8" "EPS_TRV_WAVES=8"; do
  env $kn timeout 600 python scripts/bench_graph.py --rows $ROWS --dim 768 --data uniform --L 2000 --T 4 --reps 3 --load-graph /tmp/g_ll.bin 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.readline()); print('$kn', j['config'][-11:], 'kernel_ms %.2f' % j['kernel_ms'], 'qps %.0f' % j['qps'])"
done
