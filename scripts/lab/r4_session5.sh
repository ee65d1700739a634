#!/bin/bash
# r4 GPU session 5: the single-query chain after the launch fusions (query prep + fragment copy + prologue in one launch, seed selection
# straight into the candidate lists, finalisation in the last re-rank, 3 stages for <= 4 queries): exactness tests, then latency by
# stage count and by batch size (1M x 768)
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out/r4s5
( timeout 900 python -m pytest tests/test_gpu_mfma_i8.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_sharded.py -m gpu -q 2>&1 | tail -15 ) > gpurun_out/r4s5/pytest.txt
tail -8 gpurun_out/r4s5/pytest.txt
( for st in 2 3 4; do echo -n "EPS_MFMA_STAGES=$st "; EPS_MFMA_STAGES=$st timeout 300 python scripts/prof_single_query.py 1000000 768 2>/dev/null | tail -1; done
  echo -n "default "; timeout 300 python scripts/prof_single_query.py 1000000 768 2>/dev/null | tail -1
  timeout 300 python scripts/lab/stages_by_batch.py 1000000 768 2>/dev/null | tail -1
  EPS_MFMA_STAGES=3 timeout 300 python scripts/lab/stages_by_batch.py 1000000 768 2>/dev/null | tail -1 ) > gpurun_out/r4s5/latency.txt 2>&1
cat gpurun_out/r4s5/latency.txt
