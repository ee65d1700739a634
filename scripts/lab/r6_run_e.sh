#!/bin/bash
# r6 session E: wavefronts per query where the LDS queues leave two workgroups per CU (T = 4, L = 1000 / 2000), 10M x 768 proxy
export EPS_TUNING_FROM_ENV=1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6e
mkdir -p $O
cd $R
CASES=4:2000,4:1000,8:1000,4:700
for w in 4 8 16; do
(VARIANT=waves$w EPS_TRV_WAVES=$w timeout 900 python scripts/lab/r5_trv_proxy.py 10000000 768 48 $CASES > $O/proxy_w$w.jsonl 2> $O/proxy_w$w.err)
done
(VARIANT=waves8_prof EPS_TRV_WAVES=8 EPS_TRV_PROF=1 REPS=1 timeout 900 python scripts/lab/r5_trv_proxy.py 10000000 768 48 4:2000 2>&1 | grep "eps trv" | tail -14 > $O/phase_T4_L2000_w8.txt)
(VARIANT=t1_prof EPS_TRV_PROF=1 REPS=1 timeout 900 python scripts/lab/r5_trv_proxy.py 10000000 768 48 1:2000 2>&1 | grep "eps trv" | tail -14 > $O/phase_T1_L2000.txt)
cat $O/proxy_w*.jsonl | cut -c1-120; cat $O/phase_T4_L2000_w8.txt $O/phase_T1_L2000.txt
