#!/bin/bash
# r6 session B (one gpurun call): the headline kernel's cap with operand bytes that stay random under the ablations; the embedding-like step by grid cut
export EPS_TUNING_FROM_ENV=1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6b
mkdir -p $O
cd $R
rm -f gpurun_out/r6cap/ab.txt
(timeout 1500 bash scripts/lab/r6_headline_cap.sh > $O/cap.log 2>&1)
for c in none 7 6 5 4; do
  if [ $c == none ]; then unset EPS_MIRROR_CLIP; else export EPS_MIRROR_CLIP=$c; fi
  echo "EPS_MIRROR_CLIP=$c" >> $O/embedding_clip.txt
  (timeout 600 python scripts/lab/r6_embedding_steps.py 10000000 10 2>&1 | grep "embedding-like" >> $O/embedding_clip.txt)
done
cat gpurun_out/r6cap/ab.txt; cat $O/embedding_clip.txt
