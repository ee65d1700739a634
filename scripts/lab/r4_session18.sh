#!/bin/bash
# r4 session 18: the one-pass search up to 8 queries: exactness, p50 by batch size (1M x 768) against the staged chain
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r4s18
timeout 500 python -m pytest tests/test_gpu_mfma_i8.py -m gpu -x -q -k "one_pass or one_to_four or handful or auto_builds" 2>&1 | grep -E "passed|failed|^E  " | cut -c1-400 | head -6
for v in 1 0; do
  echo "EPS_FLAT_ONE_PASS=$v $(EPS_FLAT_ONE_PASS=$v timeout 300 python scripts/lab/stages_by_batch.py 1000000 768 2>/dev/null | tail -1 | tr '\n' ' ')" | tee -a gpurun_out/r4s18/by_batch.txt
done
