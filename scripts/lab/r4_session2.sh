#!/bin/bash
# r4 GPU session 2: the whole GPU suite on the new default library (centred grid, VGPR-form accumulators, sharded device buffers, new
# full-size tests), epilogue ablations of the filter kernel, the default bench line
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out/r4s2
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > gpurun_out/r4s2/pytest.txt
bash scripts/lab/ab_libs.sh 1 vi0 vf7 vf7a1 vf7a2 vf7a4 vf7a5 > gpurun_out/r4s2/ab.txt 2>&1
( time timeout 900 python bench.py ) > gpurun_out/r4s2/bench.json 2> gpurun_out/r4s2/bench.err
tail -12 gpurun_out/r4s2/pytest.txt; cat gpurun_out/r4s2/ab.txt; cut -c1-600 gpurun_out/r4s2/bench.json; tail -5 gpurun_out/r4s2/bench.err
