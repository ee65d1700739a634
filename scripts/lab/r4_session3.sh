#!/bin/bash
# r4 GPU session 3: tile-level epilogue test A/B, the filter-kernel tests on it, then the 10M x 768 graph run (build, sweeps, contract line, rocprofv3 stats + PMC)
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out/r4s3
bash scripts/lab/ab_libs.sh 2 notile tile > gpurun_out/r4s3/ab.txt 2>&1
cp scripts/lab/_ab/tile.so vectordb_amd/lib/libepsilla_gfx950.so
( timeout 900 python -m pytest tests/test_gpu_mfma_i8.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_build.py tests/test_gpu_traverse.py -m gpu -q -k "not side_by_side" 2>&1 | tail -15 ) > gpurun_out/r4s3/pytest.txt
cat gpurun_out/r4s3/ab.txt; tail -6 gpurun_out/r4s3/pytest.txt
bash scripts/run_10m_graph_r4.sh > gpurun_out/r4s3/graph.log 2>&1
tail -5 gpurun_out/r4g/graph_10M_uniform.jsonl | cut -c1-500; cut -c1-700 gpurun_out/r4g/bench_graph_10M.json; cat gpurun_out/r4g/graph_10M_build.txt | tail -8; cat gpurun_out/r4g/graph_10M_phase_profile.txt
