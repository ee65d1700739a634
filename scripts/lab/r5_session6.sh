#!/bin/bash
# r5 session 6: whole GPU suite on the new library (tuning table, edge constants, shapes), then the default bench line
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5s6
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | cut -c1-400 | tee $O/tests.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.err
python - <<'P'
import json
j = json.loads(open("gpurun_out/r5s6/bench.json").read().strip().splitlines()[-1])
print("value", j["value"], "ms", j["ms_per_step"], "frac", j["roofline"]["frac"], "e2e", j["end_to_end"]["value"], j["end_to_end"]["frac_of_device_resident"], j["end_to_end"]["unpipelined_host_pointers"])
for k, v in j.get("configs", {}).items():
    print(k, json.dumps(v)[:600])
P
