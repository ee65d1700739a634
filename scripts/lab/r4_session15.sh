#!/bin/bash
# r4 session 15: stream8_kernel ablations (EPS_S8_ABLATE bits, see Stream8Args::ablate), three rounds
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r4s15
for r in 1 2 3; do
  for v in 0 1 4 8; do
    echo "round $r EPS_S8_ABLATE=$v $(EPS_S8_ABLATE=$v timeout 200 python scripts/prof_single_query.py 2>/dev/null | tail -1)" | tee -a gpurun_out/r4s15/latency.txt
  done
done
