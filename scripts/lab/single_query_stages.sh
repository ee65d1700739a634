#!/bin/bash
# single-query latency of the 8-bit filter path by stage count (1M x 768)
R=$GRAFT_REPO_ROOT
cd $R
for st in 2 3 4 5 6 7; do
  echo -n "EPS_MFMA_STAGES=$st "
  EPS_MFMA_STAGES=$st timeout 300 python scripts/prof_single_query.py 1000000 768 2>/dev/null | tail -1
done
