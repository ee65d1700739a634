#!/bin/bash
# r6 session F: the one-pass search on tables with folded margins - tests, then 1 / 3 / 8 / 16 queries per call on a 1M x 768 embedding-like table by grid cut
export EPS_TUNING_FROM_ENV=1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6f
mkdir -p $O
cd $R
(timeout 1500 python -m pytest tests/test_gpu_mfma_i8.py tests/test_gpu_rotated_frame.py tests/test_gpu_traverse.py -m gpu -q -x --timeout 1400 2>&1 | tail -25 > $O/tests.txt)
tail -12 $O/tests.txt
for c in none 7 6 5; do
  if [ $c == none ]; then unset EPS_MIRROR_CLIP; else export EPS_MIRROR_CLIP=$c; fi
  echo "EPS_MIRROR_CLIP=$c" >> $O/rot_one_pass_1M_by_cut.txt
  (EPS_DEBUG_ONE_PASS_OVERFLOW=0 timeout 600 python scripts/lab/r6_rotated_one_pass_probe.py 1000000 2>&1 | grep "^frame auto" >> $O/rot_one_pass_1M_by_cut.txt)
done
export EPS_MIRROR_CLIP=6
echo "EPS_MIRROR_CLIP=6 EPS_S8_FOLD=0 (staged chain)" >> $O/rot_one_pass_1M_by_cut.txt
(EPS_S8_FOLD=0 timeout 600 python scripts/lab/r6_rotated_one_pass_probe.py 1000000 2>&1 | grep "^frame auto" >> $O/rot_one_pass_1M_by_cut.txt)
cut -c1-200 $O/rot_one_pass_1M_by_cut.txt
