#!/bin/bash
# r5 session 5: edge constants (acc0 of the neighbours stored with the adjacency list): exactness, then the proxy A/B against the r4 kernel
cd ${GRAFT_REPO_ROOT:-.}
timeout 600 python -m pytest tests/test_gpu_traverse.py -m gpu -x -q -k "lockstep or invisible or outside or outlier or local_queue or visited or edge_cases or switches" 2>&1 | tail -5 | cut -c1-300
bash scripts/lab/r5_trv_ab.sh r5s5 "f0e0 ec ec_u83 ec_p6u2 ec_u4 f0e0 ec" "4:500,1:500,1:100,4:100" ec
