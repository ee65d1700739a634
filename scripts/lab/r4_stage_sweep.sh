#!/bin/bash
# r4: stage-count sweep of the batched flat scan at the headline shape with the centred grid (half the candidates per stage)
R=${GRAFT_REPO_ROOT:-.}
cd $R
for st in 3 4 5 6 7 8; do
EPS_MFMA_STAGES=$st timeout 600 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --graph-rows 0 --configs none --recall-queries 128 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('stages $st', 'ms/step %.3f' % j['ms_per_step'], 'main kernel %.3f' % j['roofline']['kernel_ms_per_launch'], 'all stages %.3f' % j['roofline']['all_stage_launches']['kernel_ms'], 'q/s %.0f' % j['value'], 'recall', j['recall_at_10'], 'rerank rows/query %.0f' % j['stats']['rerank_rows_per_query'], 'ovf', j['stats']['overflow_queries'])"
done
