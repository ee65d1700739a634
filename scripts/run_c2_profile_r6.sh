#!/bin/bash
# ONE gpurun call: BASELINE configs[1] (1M x 768, one query per call) - the one-pass search under rocprofv3: kernel stats + HBM traffic of the pass
set -x
export EPS_TUNING_FROM_ENV=1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c2
mkdir -p $O
(timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o stats -- python $R/scripts/prof_single_query.py 1000000 768 > $O/prof_stats.log 2>&1)
(timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/prof_fetch -o fetch -- python $R/scripts/prof_single_query.py 1000000 768 > $O/prof_fetch.log 2>&1)
(timeout 600 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/prof_write -o write -- python $R/scripts/prof_single_query.py 1000000 768 > $O/prof_write.log 2>&1)
cd $R
f=$(find $O/prof_stats -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f $O/kernel_stats.csv 8
f1=$(find $O/prof_fetch -name "*.db" | head -1); f2=$(find $O/prof_write -name "*.db" | head -1)
[ -n "$f1" ] && python scripts/rocpd_pmc.py $O/pmc.csv $f1 $f2 | grep -E "kernel,|stream8|rerank" | head -12
grep "p50" $O/prof_stats.log | tail -1
find $O -name "*.db" -size +20M -delete
