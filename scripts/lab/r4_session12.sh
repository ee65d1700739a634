#!/bin/bash
# r4 session 12: shader clock and socket power while the filter kernel runs, one- vs two-workgroups-per-CU form (rocm-smi sampled in the background)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r4s12
ls /sys/class/drm/ > gpurun_out/r4s12/drm.txt 2>&1
for v in 0 1; do
  ( while true; do echo "t $(date +%s.%N)"; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power\|fclk\|mclk"; sleep 0.2; done ) > gpurun_out/r4s12/smi.$v.txt 2>&1 &
  SPID=$!
  EPS_MFMA_TWO_PER_CU=$v timeout 300 python bench.py --steps 600 --warmup 5 --cpu-seconds 0 --graph-rows 0 --configs none --recall-queries 128 2> gpurun_out/r4s12/bench.$v.err | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('two_per_cu=$v', 'ms/step %.3f' % j['ms_per_step'], 'kernel %.3f' % j['roofline']['kernel_ms_per_launch'], 'frac %.4f' % j['roofline']['frac'])" | tee -a gpurun_out/r4s12/ab.txt
  kill $SPID
  wait $SPID 2>/dev/null
  echo "t_end $(date +%s.%N)" >> gpurun_out/r4s12/smi.$v.txt
done
for v in 0 1; do echo "== $v"; grep -i "sclk" gpurun_out/r4s12/smi.$v.txt | awk '{print $NF}' | sort | uniq -c | sort -rn | head -8; grep -i "power" gpurun_out/r4s12/smi.$v.txt | awk '{print $NF}' | sort -n | tail -5 | tr '\n' ' '; echo; done
