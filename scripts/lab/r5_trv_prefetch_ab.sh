#!/bin/bash
# r5 (late): step b touches the gathered neighbours' mirror rows while the visited atomics are in flight (EPS_TRV_PREFETCH=1) vs not, 10M x 768 proxy
cd ${GRAFT_REPO_ROOT:-.}
export EPS_TUNING_FROM_ENV=1
mkdir -p gpurun_out
for rep in 1 2; do
  for v in 0 1; do
    EPS_TRV_PREFETCH=$v VARIANT=prefetch$v timeout 400 python scripts/lab/r5_trv_proxy.py 10000000 768 48 1:100,4:100,4:500,1:500 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('%-10s T=%d L=%-4d %8.3f ms  %8d q/s  frac %.4f  evals %.0f  crc %d' % (j['variant'], j['T'], j['L'], j['kernel_ms'], j['qps'], j['frac_of_8TBps'], j['evals_per_query'], j['ids_crc']))"
  done
done
