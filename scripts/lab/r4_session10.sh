#!/bin/bash
# r4 session 10: the two-workgroups-per-CU form of the filter kernel (NRB = 4, EPS_MFMA_TWO_PER_CU=1): exactness, then A/B on the headline config
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r4s10
EPS_MFMA_TWO_PER_CU=1 timeout 400 python -m pytest tests/test_gpu_mfma_i8.py -m gpu -x -q -k "exact or deleted or outlier" > gpurun_out/r4s10/pytest_two.txt 2>&1
tail -5 gpurun_out/r4s10/pytest_two.txt
for r in 1 2; do
  for v in 0 1; do
    EPS_MFMA_TWO_PER_CU=$v timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --graph-rows 0 --configs none --recall-queries 128 2> gpurun_out/r4s10/bench.$v.$r.err | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('two_per_cu=$v', $r, 'ms/step %.3f' % j['ms_per_step'], 'kernel %.3f' % j['roofline']['kernel_ms_per_launch'], 'frac %.4f' % j['roofline']['frac'], 'recall', j['recall_at_10'], 'rerank', j['stats']['rerank_rows_per_query'], 'ovf', j['stats']['overflow_queries'])" | tee -a gpurun_out/r4s10/ab.txt
  done
done
