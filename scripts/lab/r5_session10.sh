#!/bin/bash
# r5 session 10: module boundary timing (query_batch phases at 10M x 768) + bench graph e2e check at 1M
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5s10
EPS_DROPIN_TIMING=1 TAG=r5s10/module_10M BATCHES=6 bash scripts/module_10m.sh 2>&1 | tail -3 | cut -c1-1500
grep "epsilla.query_batch" gpurun_out/r5s10/module_10M.log | tail -8
timeout 600 python bench.py --mode graph --rows 1000000 --steps 10 --warmup 3 --cpu-seconds 0 --configs none --power-seconds 0 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline()); print('graph 1M: value', j['value'], 'e2e', j['end_to_end']['value'], j['end_to_end']['frac_of_device_resident'], 'unpipelined', j['end_to_end']['unpipelined_host_pointers']['value'])"
