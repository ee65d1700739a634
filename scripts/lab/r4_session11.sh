#!/bin/bash
# r4 session 11: tile-phase cycle profile (-DEPS_V7_PROF) of the one- and two-workgroups-per-CU forms
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r4s11
cp vectordb_amd/lib/libepsilla_gfx950.so /tmp/cur.so
cp scripts/lab/_ab/prof.so vectordb_amd/lib/libepsilla_gfx950.so
for v in 0 1; do
  EPS_MFMA_TWO_PER_CU=$v timeout 300 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --graph-rows 0 --configs none --recall-queries 128 2> gpurun_out/r4s11/prof.$v.err > gpurun_out/r4s11/prof.$v.json
  echo "two_per_cu=$v"; grep "v7 prof" gpurun_out/r4s11/prof.$v.err | tail -3
done
cp /tmp/cur.so vectordb_amd/lib/libepsilla_gfx950.so
