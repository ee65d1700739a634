#!/bin/bash
# A/B of the traversal's 8-bit prefilter (one gpurun call): ROWS x 768 uniform, graph built once, sweeps with the prefilter off / on
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
ROWS=${1:-1000000}
mkdir -p $O
cd $R
EPS_TRV_PREFILTER=0 timeout 1200 python scripts/bench_graph.py --rows $ROWS --dim 768 --data uniform --L 500,2000 --T 1,4 --reps 3 --save-graph /tmp/g_ab.bin > $O/trv_ab_${ROWS}_off.jsonl 2> $O/trv_ab_${ROWS}_off.err
EPS_TRV_PREFILTER=1 timeout 900 python scripts/bench_graph.py --rows $ROWS --dim 768 --data uniform --L 500,2000 --T 1,4 --reps 3 --load-graph /tmp/g_ab.bin > $O/trv_ab_${ROWS}_on.jsonl 2> $O/trv_ab_${ROWS}_on.err
EPS_TRV_PROF=1 EPS_TRV_PREFILTER=0 timeout 600 python scripts/bench_graph.py --rows $ROWS --dim 768 --data uniform --L 500 --T 4 --reps 1 --load-graph /tmp/g_ab.bin 2>&1 | grep "eps trv" | tail -12 > $O/trv_ab_${ROWS}_prof_off.txt
EPS_TRV_PROF=1 EPS_TRV_PREFILTER=1 timeout 600 python scripts/bench_graph.py --rows $ROWS --dim 768 --data uniform --L 500 --T 4 --reps 1 --load-graph /tmp/g_ab.bin 2>&1 | grep "eps trv" | tail -13 > $O/trv_ab_${ROWS}_prof_on.txt
tail -5 $O/trv_ab_${ROWS}_off.jsonl $O/trv_ab_${ROWS}_on.jsonl; cat $O/trv_ab_${ROWS}_prof_off.txt $O/trv_ab_${ROWS}_prof_on.txt; tail -3 $O/trv_ab_${ROWS}_on.err
