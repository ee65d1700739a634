#!/bin/bash
# build-time A/B of the Link searches' 8-bit prefilter: ROWS x 768 uniform, stage timings from EPS_DEBUG
R=$GRAFT_REPO_ROOT
ROWS=${1:-1000000}
cd $R
for pf in 0 1; do
  echo "EPS_BUILD_PREFILTER=$pf"
  EPS_BUILD_PREFILTER=$pf EPS_DEBUG=1 timeout 1200 python scripts/build_timing.py $ROWS 768 2>&1 | grep -E "eps build|build_s|recall" | cut -c1-200
done
