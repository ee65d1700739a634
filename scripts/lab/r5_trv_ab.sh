#!/bin/bash
# r5: traversal kernel variants (scripts/lab/_ab/<name>.so, build_trv_variants.sh) on the 10M x 768 random-graph proxy.
#   usage: r5_trv_ab.sh outdir "v1 v2 ..." ["T:L,T:L"] [profile-variant ...]
cd ${GRAFT_REPO_ROOT:-.}
export EPS_TUNING_FROM_ENV=1
O=gpurun_out/$1; shift
V="$1"; shift
C="${1:-4:500,1:500,1:100,4:100}"; shift
mkdir -p $O
cp vectordb_amd/lib/libepsilla_gfx950.so /tmp/cur_lib.so
for v in $V; do
  cp scripts/lab/_ab/$v.so vectordb_amd/lib/libepsilla_gfx950.so
  VARIANT=$v timeout 400 python scripts/lab/r5_trv_proxy.py ${ROWS:-10000000} 768 48 $C 2>$O/proxy_$v.err | tee -a $O/proxy.jsonl | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('%-8s T=%d L=%-4d %8.3f ms  frac %.4f  crc %d' % (j['variant'], j['T'], j['L'], j['kernel_ms'], j['frac_of_8TBps'], j['ids_crc']))"
done
for v in "$@"; do
  cp scripts/lab/_ab/$v.so vectordb_amd/lib/libepsilla_gfx950.so
  for c in 4:500 1:100; do
    EPS_TRV_PROF=1 REPS=1 VARIANT=$v timeout 300 python scripts/lab/r5_trv_proxy.py ${ROWS:-10000000} 768 48 $c 2>&1 | grep "eps trv" | tail -13 | tee -a $O/phase_profile_$v.txt
  done
done
cp /tmp/cur_lib.so vectordb_amd/lib/libepsilla_gfx950.so
