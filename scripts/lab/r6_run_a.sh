#!/bin/bash
# r6 session A (one gpurun call): single-query traversal latency probe; the embedding-like step's kernel breakdown; the headline kernel's cap
export EPS_TUNING_FROM_ENV=1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6a
mkdir -p $O
cd $R
(timeout 600 python scripts/lab/r6_graph_latency.py > $O/graph_latency.txt 2>&1)
(EPS_DEBUG_ONE=1 timeout 600 python scripts/lab/r6_embedding_steps.py 10000000 10 > $O/embedding_steps.txt 2>&1)
cd /tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_emb -o stats -- python $R/scripts/lab/r6_embedding_steps.py 10000000 10 > $O/prof_emb.log 2>&1)
cd $R
f=$(find $O/prof_emb -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f $O/embedding_kernel_stats.csv 16
find $O -name "*.db" -delete
(timeout 1500 bash scripts/lab/r6_headline_cap.sh > $O/cap.log 2>&1)
tail -20 $O/graph_latency.txt; tail -3 $O/embedding_steps.txt; head -12 $O/embedding_kernel_stats.csv | cut -c1-160; cat gpurun_out/r6cap/ab.txt
