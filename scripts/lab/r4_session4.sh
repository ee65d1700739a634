#!/bin/bash
# r4 GPU session 4: finer epilogue ablations (what do the 16 branch pairs cost, what does the hit code cost when it never runs), the
# random row-gather ceiling, the drop-in tests (C++ batch entry, rebuild on a side index), sharded tests
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out/r4s4
bash scripts/lab/ab_libs.sh 1 notile nt_a2 nt_a8 nt_a16 > gpurun_out/r4s4/ab.txt 2>&1
cat gpurun_out/r4s4/ab.txt
( timeout 300 scripts/lab/gather_peak 30 ) > gpurun_out/r4s4/gather_peak.txt 2>&1
cat gpurun_out/r4s4/gather_peak.txt
( timeout 1200 python -m pytest tests/test_dropin.py tests/test_epsilla_module.py tests/test_gpu_sharded.py tests/test_gpu_traverse.py -m gpu -q -k "not side_by_side" 2>&1 | tail -25 ) > gpurun_out/r4s4/pytest.txt
tail -25 gpurun_out/r4s4/pytest.txt
