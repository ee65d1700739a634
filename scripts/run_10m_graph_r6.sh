#!/bin/bash
# ONE gpurun call (r6): 10M x 768 graph build + traversal sweep (8-bit prefilter on, and off for the A/B) + contract line +
# rocprofv3 stats + PMC passes (graph file in /tmp)
set -x
export EPS_TUNING_FROM_ENV=1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6g
mkdir -p $O
cd $R
(EPS_DEBUG=1 timeout 1500 python scripts/bench_graph.py --rows 10000000 --dim 768 --data uniform --L 500,2000 --T 1,4 --reps 2 --save-graph /tmp/g10m.bin > $O/graph_10M_uniform.jsonl 2> $O/graph_10M_uniform.err)
grep "eps build" $O/graph_10M_uniform.err > $O/graph_10M_build.txt
(EPS_TRV_PREFILTER=0 timeout 900 python scripts/bench_graph.py --rows 10000000 --dim 768 --data uniform --L 500 --T 1,4 --reps 2 --load-graph /tmp/g10m.bin > $O/graph_10M_uniform_prefilter_off.jsonl 2>/dev/null)
(EPS_TRV_PROF=1 timeout 900 python scripts/bench_graph.py --rows 10000000 --dim 768 --data uniform --L 500 --T 4 --reps 1 --load-graph /tmp/g10m.bin 2>&1 | grep "eps trv" | tail -13 > $O/graph_10M_phase_profile.txt)
cd /tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_graph_stats -o stats -- python $R/scripts/prof_graph.py /tmp/g10m.bin > $O/prof_graph_stats.log 2>&1)
(timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/prof_graph_fetch -o fetch -- python $R/scripts/prof_graph.py /tmp/g10m.bin > $O/prof_graph_fetch.log 2>&1)
(timeout 600 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/prof_graph_write -o write -- python $R/scripts/prof_graph.py /tmp/g10m.bin > $O/prof_graph_write.log 2>&1)
cd $R
f=$(find $O/prof_graph_stats -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f $O/graph_10M_kernel_stats.csv 12
f1=$(find $O/prof_graph_fetch -name "*.db" | head -1); f2=$(find $O/prof_graph_write -name "*.db" | head -1)
[ -n "$f1" ] && python scripts/rocpd_pmc.py $O/graph_10M_pmc.csv $f1 $f2 | grep -E "kernel,|eps" | head -40
# (the contract line LAST: its traffic_reference then names the PMC file this very run leaves - VERDICT r5 9c)
(timeout 900 python bench.py --mode graph --load-graph /tmp/g10m.bin --T 4 --L 500 --steps 5 --warmup 2 --cpu-seconds 40 --configs none > $O/bench_graph_10M.json 2> $O/bench_graph_10M.err)
find $O -name "*.db" -size +20M -delete
du -sh $O
