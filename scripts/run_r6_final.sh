#!/bin/bash
# r6, the round's closing gpurun call on the final library: the whole GPU suite, the default bench line, the 10M x 768 uniform graph sweep (build + T 1 / 4 x L 500 / 2000),
# the manifold set's build + sweep + contract line
export EPS_TUNING_FROM_ENV=1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6final
mkdir -p $O
cd $R
(timeout 1800 python -m pytest tests -m gpu -q --timeout 1500 2>&1 | tail -12 > $O/gpu_tests.txt)
unset EPS_TUNING_FROM_ENV
(timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err)
export EPS_TUNING_FROM_ENV=1
(EPS_DEBUG=1 timeout 1500 python scripts/bench_graph.py --rows 10000000 --dim 768 --data uniform --L 500,1000,2000 --T 1,4 --reps 2 > $O/graph_10M_uniform.jsonl 2> $O/graph_10M_uniform.err)
grep "eps build" $O/graph_10M_uniform.err > $O/graph_10M_build.txt
rm -f $O/graph_10M_uniform.err
if [ -z "$NO_MANIFOLD" ]; then
(timeout 1500 bash scripts/run_10m_manifold_r6.sh > $O/manifold.log 2>&1)
fi
tail -4 $O/gpu_tests.txt; cut -c1-330 $O/graph_10M_uniform.jsonl
