#!/bin/bash
# r6, VERDICT r5 #2 "or prove the cap": the headline launch (mfma_filter_kernel_v7<2, FM_IDS, int8>, 7.28M rows x 1024 queries) next to its own
# ablations on ONE box, alternating, with the shader clock and socket power of each under sustained load:
#   cur       the product kernel
#   noepi     no epilogue work (ABL 1)
#   nolds     no epilogue, no LDS fragment reads (ABL 1|32): MFMAs + LDS-DMA ring + query-fragment loads
#   noglobal  no epilogue, no global traffic in the K loop (ABL 1|64): MFMAs + LDS fragment reads
#   mfmaonly  ABL 1|32|64: the kernel's MFMA stream (random operand bytes), barriers and scalar code - nothing else
# (variants: scripts/lab/build_variants.sh, built on the CPU container; answers of the ablated ones are wrong by construction)
R=${GRAFT_REPO_ROOT:-.}
cd $R
O=gpurun_out/r6cap
mkdir -p $O
cp vectordb_amd/lib/libepsilla_gfx950.so /tmp/cur.so
for r in 1 2; do
  for v in cur noepi nolds noglobal mfmaonly; do
    cp scripts/lab/_ab/$v.so vectordb_amd/lib/libepsilla_gfx950.so
    timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --graph-rows 0 --configs none --recall-queries 64 --power-seconds 4 --no-e2e --no-pmc 2> $O/$v.$r.err | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
p = j['roofline'].get('under_load', {}) or {}
print('$v', $r, 'ms/step %.3f' % j['ms_per_step'], 'kernel %.3f' % j['roofline']['kernel_ms_per_launch'], 'frac %.4f' % j['roofline']['frac'], 'recall', j['recall_at_10'],
      'sclk', p.get('sclk_mhz_under_load'), 'W', p.get('socket_power_w_under_load'), 'sustained ms/step', p.get('ms_per_step_sustained'))" | tee -a $O/ab.txt
  done
done
cp /tmp/cur.so vectordb_amd/lib/libepsilla_gfx950.so
# the instruction alone, same board (scripts/lab/mfma_peak_i8.hip: v_mfma_i32_32x32x32_i8, 1 wave per SIMD, random bytes / zeros)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak_i8 scripts/lab/mfma_peak_i8.hip 2>/dev/null && /tmp/mfma_peak_i8 | tee $O/mfma_peak_i8.txt
