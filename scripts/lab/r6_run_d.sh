#!/bin/bash
# r6 session D: worker queues laid out with their occupancy bound (T = 4, L = 2000 back in LDS) - tests, then the 10M x 768 proxy A/B; the one-pass probe on a 1M rotated table
export EPS_TUNING_FROM_ENV=1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6d
mkdir -p $O
cd $R
(timeout 1500 python -m pytest tests/test_gpu_traverse.py tests/test_gpu_rotated_frame.py tests/test_gpu_parity.py tests/test_golden.py -m gpu -q -x --timeout 1400 2>&1 | tail -15 > $O/tests.txt)
tail -3 $O/tests.txt
CASES=4:2000,1:2000,4:1000,1:1000,4:500,1:500,8:1000
(VARIANT=lds_bound timeout 900 python scripts/lab/r5_trv_proxy.py 10000000 768 48 $CASES > $O/proxy_bound.jsonl 2> $O/proxy_bound.err)
(VARIANT=hbm_queues EPS_TRV_LDS_KB=40 timeout 900 python scripts/lab/r5_trv_proxy.py 10000000 768 48 4:2000,4:1000,8:1000 > $O/proxy_hbm.jsonl 2> $O/proxy_hbm.err)
cat $O/proxy_bound.jsonl $O/proxy_hbm.jsonl | cut -c1-330
(timeout 600 python scripts/lab/r6_rotated_one_pass_probe.py 1000000 2>&1 | grep -v amdgpu.ids > $O/rot_one_pass_1M.txt)
cut -c1-200 $O/rot_one_pass_1M.txt | head -10
