#!/bin/bash
# r6 session G: the one-pass search of configs[1] under rocprofv3 (kernel stats + PMC); the embedding-like step by stage count
export EPS_TUNING_FROM_ENV=1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6g2
mkdir -p $O
cd $R
(timeout 1200 bash scripts/run_c2_profile_r6.sh > $O/c2_profile.log 2>&1)
for st in 5 6 7 8; do
  echo "EPS_MFMA_STAGES=$st" >> $O/embedding_stages.txt
  (EPS_MFMA_STAGES=$st timeout 600 python scripts/lab/r6_embedding_steps.py 10000000 10 2>&1 | grep "embedding-like" >> $O/embedding_stages.txt)
done
cat $O/embedding_stages.txt | cut -c1-200; tail -12 $O/c2_profile.log | cut -c1-200
