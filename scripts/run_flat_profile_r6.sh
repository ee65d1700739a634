#!/bin/bash
# r6, ONE gpurun call (PMC and kernel-stats passes FIRST, the bench line last): the headline line (10M x 768, batch 1024, exact flat scan) + rocprofv3 kernel stats + PMC passes of the same command
set -x
export EPS_TUNING_FROM_ENV=1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6flat
mkdir -p $O
cd $R
STEPS=${STEPS:-20}
cd /tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o stats -- python $R/bench.py --steps $STEPS --warmup 5 --cpu-seconds 0 --graph-rows 0 --configs none --recall-queries 64 --no-pmc > $O/prof_stats.log 2>&1)
if [ -z "$NOPMC" ]; then
(timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/prof_fetch -o fetch -- python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --graph-rows 0 --configs none --recall-queries 64 --no-pmc > $O/prof_fetch.log 2>&1)
(timeout 600 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/prof_write -o write -- python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --graph-rows 0 --configs none --recall-queries 64 --no-pmc > $O/prof_write.log 2>&1)
(timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/prof_sq -o sq -- python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --graph-rows 0 --configs none --recall-queries 64 --no-pmc > $O/prof_sq.log 2>&1)
fi
cd $R
f=$(find $O/prof_stats -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f $O/kernel_stats.csv 8
f1=$(find $O/prof_fetch -name "*.db" | head -1); f2=$(find $O/prof_write -name "*.db" | head -1); f3=$(find $O/prof_sq -name "*.db" | head -1)
[ -n "$f1" ] && python scripts/rocpd_pmc.py $O/pmc.csv $f1 $f2 $f3 | grep -E "kernel,|mfma_filter|rerank" | head -40
cd $R
(timeout 900 python bench.py --steps $STEPS --warmup 5 > $O/bench.json 2> $O/bench.err)
find $O -name "*.db" -size +20M -delete
du -sh $O
