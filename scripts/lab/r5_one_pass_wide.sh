#!/bin/bash
# r5 (late): k = 17..64 (128 table slots) and compiled filter programs (mask launch) on the one-pass form: the tests, then p50 against the staged chain
mkdir -p gpurun_out
export EPS_TUNING_FROM_ENV=1
{
  timeout 600 python -m pytest tests/test_gpu_mfma_i8.py -m gpu -x -q 2>&1 | tail -15
  timeout 300 python scripts/lab/one_pass_wide_k_and_programs.py 2>&1 | tail -40
} > gpurun_out/r5_one_pass_wide.txt 2>&1
cat gpurun_out/r5_one_pass_wide.txt
