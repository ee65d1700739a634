#!/bin/bash
# ONE gpurun call: the low-intrinsic-dimension set at 10M x 768 (16-d uniform latent embedded linearly + 1 % noise; NOT the BASELINE recipe):
# device build, traversal sweep, bench.py --mode graph contract line with the reference's executor pool on the SAME graph
set -x
export EPS_TUNING_FROM_ENV=1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6m
mkdir -p $O
cd $R
(EPS_DEBUG=1 timeout 1500 python scripts/bench_graph.py --rows 10000000 --dim 768 --data manifold --L 50,100,200,500 --T 1,4 --reps 2 --save-graph /tmp/gm10m.bin > $O/graph_10M_manifold.jsonl 2> $O/graph_10M_manifold.err)
grep "eps build" $O/graph_10M_manifold.err > $O/graph_10M_manifold_build.txt
(EPS_TRV_PREFILTER=0 timeout 900 python scripts/bench_graph.py --rows 10000000 --dim 768 --data manifold --L 100 --T 4 --reps 2 --load-graph /tmp/gm10m.bin > $O/graph_10M_manifold_prefilter_off.jsonl 2>/dev/null)
(timeout 900 python bench.py --mode graph --data manifold --load-graph /tmp/gm10m.bin --T 4 --L 100 --steps 10 --warmup 3 --cpu-seconds 40 --configs none > $O/bench_graph_10M_manifold.json 2> $O/bench_graph_10M_manifold.err)
du -sh $O
