#!/bin/bash
# r5 session 9: visited set by generation stamp: exactness, then bitmap vs stamps on the proxy (same library, EPS_TRV_VISITED)
cd ${GRAFT_REPO_ROOT:-.}
export EPS_TUNING_FROM_ENV=1
timeout 900 python -m pytest tests/test_gpu_traverse.py -m gpu -x -q 2>&1 | tail -5 | cut -c1-300
O=gpurun_out/r5s9; mkdir -p $O
for v in bitmap stamps bitmap stamps; do
  EPS_TRV_VISITED=$v VARIANT=$v timeout 400 python scripts/lab/r5_trv_proxy.py 10000000 768 48 "4:500,1:500,1:100,4:100" 2>$O/proxy_$v.err | tee -a $O/proxy.jsonl | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('%-8s T=%d L=%-4d %8.3f ms  frac %.4f  crc %d' % (j['variant'], j['T'], j['L'], j['kernel_ms'], j['frac_of_8TBps'], j['ids_crc']))"
done
