#!/bin/bash
# r4 session 8: exactness of the early-init / flush-flag build, then A/B of the four variants on the headline config
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r4s8
timeout 420 python -m pytest tests/test_gpu_mfma_i8.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r4s8/pytest.txt 2>&1
tail -3 gpurun_out/r4s8/pytest.txt
bash scripts/lab/ab_libs.sh 2 base ei ff both 2>&1 | tee gpurun_out/r4s8/ab.txt
