#!/bin/bash
# r5 session 1: the fused distance phase + early adjacency fetch of traverse2_kernel: exactness (traversal tests), then kernel variants on the
# 10M x 768 random-graph proxy (T = 4 / 1, L = 500 / 100), then the phase profile of the default build
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5s1
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_traverse.py -m gpu -x -q -k "lockstep or invisible or outside or outlier or local_queue or visited or edge_cases" 2>&1 | tail -15 | cut -c1-300 | tee $O/tests.txt
cp vectordb_amd/lib/libepsilla_gfx950.so /tmp/cur_lib.so
for v in f0e0 cur f1e0 f0e1 f0e0u2 ud2 cur; do
  cp scripts/lab/_ab/$v.so vectordb_amd/lib/libepsilla_gfx950.so
  VARIANT=$v timeout 400 python scripts/lab/r5_trv_proxy.py 10000000 768 48 2>$O/proxy_$v.err | tee -a $O/proxy.jsonl | cut -c1-330
done
cp scripts/lab/_ab/cur.so vectordb_amd/lib/libepsilla_gfx950.so
for c in 4:500 1:500 1:100; do
  EPS_TRV_PROF=1 REPS=1 VARIANT=cur timeout 300 python scripts/lab/r5_trv_proxy.py 10000000 768 48 $c 2>&1 | grep "eps trv" | tail -13 | tee -a $O/phase_profile_cur.txt
done
cp scripts/lab/_ab/f0e0.so vectordb_amd/lib/libepsilla_gfx950.so
for c in 4:500; do
  EPS_TRV_PROF=1 REPS=1 VARIANT=f0e0 timeout 300 python scripts/lab/r5_trv_proxy.py 10000000 768 48 $c 2>&1 | grep "eps trv" | tail -13 | tee -a $O/phase_profile_f0e0.txt
done
cp /tmp/cur_lib.so vectordb_amd/lib/libepsilla_gfx950.so
