#!/bin/bash
# usage: env_sweep.sh VAR "v1 v2 ..." [reps]: bench.py (flat headline, no CPU legs) per value of an environment knob, interleaved
R=${GRAFT_REPO_ROOT:-.}; cd $R
for rep in $(seq 1 ${3:-2}); do for v in $2; do
  env $1=$v timeout 300 python bench.py --cpu-seconds 0 --graph-rows 0 --recall-queries 128 --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('$1=$v', 'ms/step %.3f' % j['ms_per_step'], 'kernel %.3f' % j['roofline']['kernel_ms_per_launch'], 'frac %.4f' % j['roofline']['frac'], 'recall', j['recall_at_10'], 'rerank %.1f' % j['stats']['rerank_rows_per_query'])"
done; done
