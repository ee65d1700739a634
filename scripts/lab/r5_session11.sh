#!/bin/bash
# r5 session 11: query_batch(as_arrays=True): test + 10M x 768 through the module
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5s11
timeout 600 python -m pytest tests/test_epsilla_module.py -m gpu -x -q 2>&1 | tail -4 | cut -c1-400
EPS_DROPIN_TIMING=1 TAG=r5s11/module_10M BATCHES=6 bash scripts/module_10m.sh 2>&1 | tail -3 | cut -c1-2500
grep "epsilla.query_batch" gpurun_out/r5s11/module_10M.log | tail -4
