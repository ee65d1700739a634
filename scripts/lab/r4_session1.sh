#!/bin/bash
# r4 GPU session 1: GPU suite on the centred 8-bit grid, then whole-library A/Bs of the headline line (r3 / centred / VGPR-form accumulators), then the manifold set
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out/r4s1
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r4s1/pytest.txt
bash scripts/lab/ab_libs.sh 2 r3 base vf6 vf7 > gpurun_out/r4s1/ab.txt 2>&1
for v in base vf7; do
  cp scripts/lab/_ab/$v.so vectordb_amd/lib/libepsilla_gfx950.so
  timeout 600 python bench.py --data manifold --steps 8 --warmup 3 --cpu-seconds 0 --graph-rows 0 --configs none --recall-queries 256 2> gpurun_out/r4s1/manifold_$v.err > gpurun_out/r4s1/manifold_$v.json
done
cp scripts/lab/_ab/base.so vectordb_amd/lib/libepsilla_gfx950.so
tail -5 gpurun_out/r4s1/pytest.txt; cat gpurun_out/r4s1/ab.txt; cut -c1-400 gpurun_out/r4s1/manifold_*.json
