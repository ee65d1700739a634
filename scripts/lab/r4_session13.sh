#!/bin/bash
# r4 GPU session 13: the manifold set at 10M x 768 on the graph path (centred prefilter), then BASELINE configs[2] through the bindings (query_batch -> epsdrop::SearchBatch, rebuild on a side index)
R=${GRAFT_REPO_ROOT:-.}
cd $R
bash scripts/run_10m_manifold_r4.sh > gpurun_out/r4m.log 2>&1
cut -c1-330 gpurun_out/r4m/graph_10M_manifold.jsonl; cut -c1-330 gpurun_out/r4m/graph_10M_manifold_prefilter_off.jsonl; cut -c1-500 gpurun_out/r4m/bench_graph_10M_manifold.json; tail -8 gpurun_out/r4m/graph_10M_manifold_build.txt
TAG=r4_module_10M EPS_MODULE_REBUILD=1 bash scripts/module_10m.sh
