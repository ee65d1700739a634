#!/bin/bash
# r5: the one-pass search for 5-8 queries (stream8_kernel<P, 8>: two rows per step, one workgroup per CU) against the staged chain (EPS_S8_MAX_Q=4),
# and the lab variant with four rows per step (scripts/lab/_ab/u8_4.so: build_variants.sh u8_4="-DEPS_S8_U8=4")
mkdir -p gpurun_out
export EPS_TUNING_FROM_ENV=1
{
  echo "chain for 5+ (EPS_S8_MAX_Q=4): $(EPS_S8_MAX_Q=4 python scripts/lab/stages_by_batch.py 1000000 768 2>&1 | tail -1)"
  echo "one pass up to 8, U=2, 1 WG/CU: $(python scripts/lab/stages_by_batch.py 1000000 768 2>&1 | tail -1)"
  echo "one pass up to 8, U=2, 2 WG/CU: $(EPS_S8_WG_PER_CU=2 python scripts/lab/stages_by_batch.py 1000000 768 2>&1 | tail -1)"
  cp vectordb_amd/lib/libepsilla_gfx950.so /tmp/cur.so
  cp scripts/lab/_ab/u8_4.so vectordb_amd/lib/libepsilla_gfx950.so
  echo "one pass up to 8, U=4, 1 WG/CU: $(python scripts/lab/stages_by_batch.py 1000000 768 2>&1 | tail -1)"
  cp /tmp/cur.so vectordb_amd/lib/libepsilla_gfx950.so
  timeout 900 python -m pytest tests/test_gpu_mfma_i8.py -m gpu -x -q -k "one_pass or one_to_eight or handful" 2>&1 | tail -5
} > gpurun_out/r5_one_pass_8q.txt 2>&1
cat gpurun_out/r5_one_pass_8q.txt
