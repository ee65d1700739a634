#!/bin/bash
# rocprofv3 kernel stats of one device build (ROWS x 768 uniform)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3b
ROWS=${1:-1000000}
mkdir -p $O
(EPS_DEBUG=1 timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o build -- python $R/scripts/build_timing.py $ROWS 768 > $O/build.log 2>&1)
cd $R
f=$(find $O/prof -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f $O/build_${ROWS}_kernel_stats.csv 25
grep "eps build" $O/build.log | cut -c1-160
find $O -name "*.db" -size +20M -delete
