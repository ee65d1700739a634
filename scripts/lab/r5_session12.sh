#!/bin/bash
# r5 session 12: DPP row sums in the traversal's / the build searches' 8-bit prefilter: exactness, proxy A/B
cd ${GRAFT_REPO_ROOT:-.}
export EPS_TUNING_FROM_ENV=1
timeout 900 python -m pytest tests/test_gpu_traverse.py tests/test_gpu_build.py -m gpu -x -q 2>&1 | tail -4 | cut -c1-300
bash scripts/lab/r5_trv_ab.sh r5s12 "nodpp dpp nodpp dpp" "4:500,1:500,1:100,4:100"
