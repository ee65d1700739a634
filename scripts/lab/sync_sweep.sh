#!/bin/bash
# rendezvous period sweep of the filter kernel: time (bench.py) and HBM fetch bytes (rocprofv3 --pmc FETCH_SIZE) per EPS_MFMA_SYNC_SHIFT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sync; mkdir -p $O
B="python $R/bench.py --cpu-seconds 0 --graph-rows 0 --recall-queries 64"
for rep in 1 2; do for sh in ${SHIFTS:-0 1 2 3}; do
  EPS_MFMA_SYNC_SHIFT=$sh $B --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('shift $sh', 'ms/step %.3f' % j['ms_per_step'], 'kernel %.3f' % j['roofline']['kernel_ms_per_launch'], 'frac %.4f' % j['roofline']['frac'])"
done; done
for sh in ${PMC_SHIFTS:-0 2 3}; do
  EPS_MFMA_SYNC_SHIFT=$sh timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/f$sh -o fetch -- $B --steps 2 --warmup 1 > $O/f$sh.log 2>&1
  f=$(find $O/f$sh -name "*.db" | head -1)
  python - $f $sh <<'P'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
q = "select dispatch_id, kernel_name, sum(value), max(duration) from counters_collection where counter_name='FETCH_SIZE' group by dispatch_id order by dispatch_id"
rows = [r for r in c.execute(q) if 'mfma_filter' in r[1]]
big = sorted(rows, key=lambda r: -r[2])[:3]
for r in big: print('shift', sys.argv[2], 'FETCH_SIZE x2 = %.3f GB' % (r[2] * 1024 * 2 / 1e9), 'dur %.2f ms' % (r[3] / 1e6))
P
done
find $O -name "*.db" -delete
