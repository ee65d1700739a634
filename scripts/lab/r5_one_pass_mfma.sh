#!/bin/bash
# r5 (late): 5..16 queries in ONE pass on the matrix cores (stream8m_kernel) against the staged chain (EPS_S8_MAX_Q=4): p50 by queries per call, then the tests
mkdir -p gpurun_out
export EPS_TUNING_FROM_ENV=1
{
  for rep in 1 2; do
    echo "chain for 5+ (EPS_S8_MAX_Q=4): $(EPS_S8_MAX_Q=4 python scripts/lab/stages_by_batch.py 1000000 768 2>&1 | tail -1)"
    echo "one pass up to 16 (default):   $(python scripts/lab/stages_by_batch.py 1000000 768 2>&1 | tail -1)"
  done
  timeout 900 python -m pytest tests/test_gpu_mfma_i8.py -m gpu -x -q -k "one_pass or sixteen or handful or call_forms" 2>&1 | tail -5
} > gpurun_out/r5_one_pass_mfma.txt 2>&1
cat gpurun_out/r5_one_pass_mfma.txt
