#!/bin/bash
# r4 GPU session 11: split re-rank + 16-wavefront seed selection for a handful of queries: exactness, latency
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out/r4s11
( timeout 900 python -m pytest tests/test_gpu_mfma_i8.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -12 ) > gpurun_out/r4s11/pytest.txt
tail -8 gpurun_out/r4s11/pytest.txt
( for sp in 1 0; do echo -n "EPS_RERANK_SPLIT=$sp "; EPS_RERANK_SPLIT=$sp timeout 300 python scripts/prof_single_query.py 1000000 768 2>/dev/null | tail -1; done
  timeout 300 python scripts/lab/stages_by_batch.py 1000000 768 2>/dev/null | tail -1
  for st in 2 3; do echo -n "EPS_MFMA_STAGES=$st "; EPS_MFMA_STAGES=$st timeout 300 python scripts/prof_single_query.py 1000000 768 2>/dev/null | tail -1; done ) > gpurun_out/r4s11/latency.txt 2>&1
cat gpurun_out/r4s11/latency.txt
