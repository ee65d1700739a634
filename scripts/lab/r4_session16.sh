#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r4s16
timeout 500 python -m pytest tests/test_gpu_mfma_i8.py -m gpu -x -q -k "one_pass or one_to_four or handful" 2>&1 | tail -2
EPS_DEBUG=1 timeout 200 python scripts/prof_single_query.py 2>&1 | grep "one pass" | tail -3 | tee gpurun_out/r4s16/debug.txt
for r in 1 2; do
  for v in 0 1; do for w in 2 4; do
    echo "round $r EPS_S8_ABLATE=$v WG=$w $(EPS_S8_WG_PER_CU=$w EPS_S8_ABLATE=$v timeout 200 python scripts/prof_single_query.py 2>/dev/null | tail -1)" | tee -a gpurun_out/r4s16/latency.txt
  done; done
done
