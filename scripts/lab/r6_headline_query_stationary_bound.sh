#!/bin/bash
# r6: what a query-stationary tile could reach at best: the headline launch with the query-fragment loads compiled out (EPS_V7_ABL 128: the row
# operand's LDS-DMA ring, the LDS fragment reads and the MFMAs stay; query fragments hold random bytes), with and without the epilogue, next to the product
# kernel and its no-epilogue form - libraries alternating on ONE box (variants: scripts/lab/build_variants.sh; answers of ablated variants are wrong by construction)
R=${GRAFT_REPO_ROOT:-.}
cd $R
O=gpurun_out/r6cap2
mkdir -p $O
rm -f $O/ab.txt
cp vectordb_amd/lib/libepsilla_gfx950.so /tmp/cur.so
for r in 1 2; do
  for v in cur noepi noqueries_noepi noqueries; do
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
