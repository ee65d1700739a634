#!/bin/bash
# r4 session 14: where the one-pass stream kernel's time goes: bare stream (EPS_S8_ABLATE=1), no periodic refresh (2), candidate counts (EPS_DEBUG)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r4s14
EPS_DEBUG=1 timeout 200 python scripts/prof_single_query.py 2>&1 | grep "one pass" | tail -3 | tee gpurun_out/r4s14/debug.txt
for v in 0 1 2; do
  for w in 2 4; do
    echo "EPS_S8_ABLATE=$v WG_PER_CU=$w $(EPS_S8_WG_PER_CU=$w EPS_S8_ABLATE=$v timeout 200 python scripts/prof_single_query.py 2>/dev/null | tail -1)" | tee -a gpurun_out/r4s14/latency.txt
  done
done
