cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6final2
(timeout 1800 python -m pytest tests -m gpu -q --timeout 1500 2>&1 | tail -8 > gpurun_out/r6final2/gpu_tests.txt)
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6final2/smoke.txt 2>&1)
(timeout 1200 python bench.py > gpurun_out/r6final2/bench_default.json 2> gpurun_out/r6final2/bench_default.err)
tail -3 gpurun_out/r6final2/gpu_tests.txt; tail -1 gpurun_out/r6final2/smoke.txt; python -c "
import json; j=json.loads(open('gpurun_out/r6final2/bench_default.json').readline()); print(j['value'], j['roofline']['frac'], j['configs']['embedding_like_10Mx768']['qps'], j['configs']['c2_1Mx768_b1_latency']['value'])"
