#!/bin/bash
# r4 GPU session 6: hit-mask hit path (per-lane bit mask instead of 16 exec-masked branches) A/B, with and without the tile-level test;
# exactness tests of every mode of the kernel (ids, approximate keys = the build's kNN stage, dense); build timing at 1M; single-query latency
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out/r4s6
bash scripts/lab/ab_libs.sh 2 nohm hm hm_tile > gpurun_out/r4s6/ab.txt 2>&1
cat gpurun_out/r4s6/ab.txt
cp scripts/lab/_ab/hm.so vectordb_amd/lib/libepsilla_gfx950.so
( timeout 1200 python -m pytest tests/test_gpu_mfma_i8.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_build.py -m gpu -q 2>&1 | tail -12 ) > gpurun_out/r4s6/pytest.txt
tail -6 gpurun_out/r4s6/pytest.txt
for v in nohm hm; do
  cp scripts/lab/_ab/$v.so vectordb_amd/lib/libepsilla_gfx950.so
  echo "== $v: 1M x 768 build"; EPS_DEBUG=1 timeout 600 python scripts/build_timing.py 1000000 768 2>&1 | grep -E "kNN graph|Link \(|build_s|seconds" | cut -c1-160
done > gpurun_out/r4s6/build_1M.txt 2>&1
cat gpurun_out/r4s6/build_1M.txt
cp scripts/lab/_ab/hm.so vectordb_amd/lib/libepsilla_gfx950.so
( echo -n "default "; timeout 300 python scripts/prof_single_query.py 1000000 768 2>/dev/null | tail -1; timeout 300 python scripts/lab/stages_by_batch.py 1000000 768 2>/dev/null | tail -1 ) > gpurun_out/r4s6/latency.txt 2>&1
cat gpurun_out/r4s6/latency.txt
