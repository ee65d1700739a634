#!/bin/bash
# r4 GPU session 7: clipped grid + per-row margins folded per batch: whole GPU suite, headline A/B against the library before it, latency, bench line
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out/r4s7
( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > gpurun_out/r4s7/pytest.txt
tail -15 gpurun_out/r4s7/pytest.txt
bash scripts/lab/ab_libs.sh 2 hm fold > gpurun_out/r4s7/ab.txt 2>&1
cat gpurun_out/r4s7/ab.txt
( echo -n "default "; timeout 300 python scripts/prof_single_query.py 1000000 768 2>/dev/null | tail -1; timeout 300 python scripts/lab/stages_by_batch.py 1000000 768 2>/dev/null | tail -1 ) > gpurun_out/r4s7/latency.txt 2>&1
cat gpurun_out/r4s7/latency.txt
( time timeout 900 python bench.py ) > gpurun_out/r4s7/bench.json 2> gpurun_out/r4s7/bench.err
cut -c1-400 gpurun_out/r4s7/bench.json; tail -4 gpurun_out/r4s7/bench.err
