#!/bin/bash
# r4 session 13: the one-pass search of a handful of queries (stream8_kernel.hpp): exactness tests, then single-query latency at 1M x 768
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r4s13
timeout 500 python -m pytest tests/test_gpu_mfma_i8.py -m gpu -x -q -k "one_pass or one_to_four or handful" > gpurun_out/r4s13/pytest.txt 2>&1
tail -15 gpurun_out/r4s13/pytest.txt
for v in 1 0; do
  echo "EPS_FLAT_ONE_PASS=$v $(EPS_FLAT_ONE_PASS=$v timeout 200 python scripts/prof_single_query.py 2>/dev/null | tail -1)" | tee -a gpurun_out/r4s13/latency.txt
done
for w in 1 4; do
  echo "EPS_S8_WG_PER_CU=$w $(EPS_S8_WG_PER_CU=$w timeout 200 python scripts/prof_single_query.py 2>/dev/null | tail -1)" | tee -a gpurun_out/r4s13/latency.txt
done
