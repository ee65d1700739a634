#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4sq
mkdir -p $O
(timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o sq -- python $R/scripts/prof_single_query.py 1000000 768 > $O/run.log 2>&1)
cd $R
f=$(find $O/prof -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f $O/single_query_kernel_stats.csv 5
python scripts/lab/single_query_timeline.py $O/single_query_kernel_stats_dispatches.csv | tee $O/single_query_timeline.txt
tail -2 $O/run.log
find $O -name "*.db" -size +20M -delete
