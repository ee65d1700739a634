#!/bin/bash
# r5 session 7: LDS-only step barriers + read-ahead of the next step's adjacency lists: exactness, then the proxy A/B
cd ${GRAFT_REPO_ROOT:-.}
export EPS_TUNING_FROM_ENV=1
timeout 600 python -m pytest tests/test_gpu_traverse.py -m gpu -x -q -k "lockstep or invisible or outside or outlier or local_queue or visited or edge_cases or switches" 2>&1 | tail -5 | cut -c1-300
bash scripts/lab/r5_trv_ab.sh r5s7 "e0full early e0 efull e0full early" "4:500,1:500,1:100,4:100" early
