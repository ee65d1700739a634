#!/bin/bash
# r5, the last call: the default bench line on the final library, then the GPU test files the earlier calls of this library did not run
export EPS_TUNING_FROM_ENV=1
O=gpurun_out/r5last
mkdir -p $O
(timeout 125 python bench.py > $O/bench_default.json 2> $O/bench_default.err)
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r5last/bench_default.json") if l.startswith("{")][-1])
    c2 = d.get("configs", {}).get("c2_1Mx768_b1_latency", {})
    print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "e2e", (d.get("end_to_end") or {}).get("value"))
    print("c2", json.dumps(c2.get("value")), json.dumps(c2.get("one_pass_widened"))[:900])
except Exception as e:
    print("bench line unreadable:", repr(e))
PY
(timeout 75 python -m pytest tests/test_gpu_traverse.py tests/test_gpu_fuzz.py tests/test_gpu_build.py -m gpu -q 2>&1 | tail -4) > $O/other_gpu_tests.txt 2>&1
cat $O/other_gpu_tests.txt
