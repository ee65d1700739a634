#!/bin/bash
# r4 session 17: regression of the flat path after the one-pass search + fused selection, then the default bench line
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r4s17
timeout 900 python -m pytest tests/test_gpu_mfma_i8.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_sharded.py tests/test_dropin.py tests/test_bench_contract.py -m gpu -x -q > gpurun_out/r4s17/pytest.txt 2>&1
tail -4 gpurun_out/r4s17/pytest.txt
timeout 600 python bench.py > gpurun_out/r4s17/bench.json 2> gpurun_out/r4s17/bench.err
python - <<PY
import json
j = json.loads(open("gpurun_out/r4s17/bench.json").readline())
print("value", j["value"], "ms/step", j["ms_per_step"], "frac", j["roofline"]["frac"])
c2 = j["configs"]["c2_1Mx768_b1_latency"]
print("c2 value", c2["value"])
for k, v in c2["gpu"].items():
    print(" ", k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk != "roofline"}, v.get("roofline", {}).get("frac"))
PY
