#!/bin/bash
# r5 (late): the one-pass call's own re-rank (s8_rerank_kernel) through every caller of the one-pass form outside tests/test_gpu_mfma_i8.py
export EPS_TUNING_FROM_ENV=1
mkdir -p gpurun_out
{
  timeout 60 python -m pytest tests/test_gpu_mfma_i8.py -m gpu -q -k "not_a_multiple" 2>&1 | tail -3
  timeout 30 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
  timeout 90 python -m pytest tests/test_gpu_full_size.py -m gpu -q -k "configs1" 2>&1 | tail -3
  timeout 150 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_dropin.py tests/test_epsilla_module.py -m gpu -q --durations=6 2>&1 | tail -14
} > gpurun_out/r5_s8_rerank_verify.txt 2>&1
cat gpurun_out/r5_s8_rerank_verify.txt
