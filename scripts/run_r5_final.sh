#!/bin/bash
# ONE gpurun call at the end of r5: the whole GPU suite, the default bench line (what the driver runs), and the widened one-pass forms under rocprofv3
export EPS_TUNING_FROM_ENV=1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5final
mkdir -p $O
cd $R
(timeout 800 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -25) > $O/gpu_tests.txt 2>&1
tail -4 $O/gpu_tests.txt
(timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err); tail -c 600 $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
(timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o stats -- python $R/scripts/lab/one_pass_wide_k_and_programs.py > $O/prof_stats.log 2>&1)
cd $R
f=$(find $O/prof_stats -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f $O/one_pass_wide_kernel_stats.csv 4
find $O -name "*.db" -delete
python - <<'PY'
import json, os
p = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r5final/bench_default.json")
try:
    d = json.loads([l for l in open(p) if l.startswith("{")][-1])
    c2 = d.get("configs", {}).get("c2_1Mx768_b1_latency", {})
    print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "e2e", (d.get("end_to_end") or {}).get("value"))
    print("c2", json.dumps(c2.get("value")), json.dumps(c2.get("one_pass_widened"))[:1500])
    print("c2 mfma_i8", json.dumps({k: v for k, v in c2.get("gpu", {}).get("mfma_i8", {}).items() if k != "roofline"}))
except Exception as e:
    print("bench line unreadable:", repr(e))
PY
