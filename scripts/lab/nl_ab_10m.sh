#!/bin/bash
# pieces-in-flight A/B at full table size on the random-graph proxy (10M x 768, deg 48, batch 1024, L = 500)
R=$GRAFT_REPO_ROOT
cd $R
cp vectordb_amd/lib/libepsilla_gfx950.so /tmp/cur.so
for v in "$@"; do
  cp scripts/lab/_ab/$v.so vectordb_amd/lib/libepsilla_gfx950.so
  WAVES=auto timeout 900 python scripts/lab/bench_random_graph.py 10000000 768 48 1024 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('$v', 'T=%d' % j['T'], 'kernel_ms %.3f' % j['kernel_ms'], 'evals %.0f' % j['evals_per_query'])"
done
cp /tmp/cur.so vectordb_amd/lib/libepsilla_gfx950.so
