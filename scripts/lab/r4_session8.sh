#!/bin/bash
# r4 GPU session 8: fold only where rows differ: the two tests that failed, the 8-bit / traversal / build suites, headline A/B
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out/r4s8
( timeout 1500 python -m pytest tests/test_gpu_mfma_i8.py tests/test_gpu_parity.py tests/test_gpu_traverse.py tests/test_gpu_build.py tests/test_gpu_fuzz.py -m gpu -q -k "not side_by_side" 2>&1 | tail -30 ) > gpurun_out/r4s8/pytest.txt
tail -12 gpurun_out/r4s8/pytest.txt
bash scripts/lab/ab_libs.sh 2 hm fold2 > gpurun_out/r4s8/ab.txt 2>&1
cat gpurun_out/r4s8/ab.txt
