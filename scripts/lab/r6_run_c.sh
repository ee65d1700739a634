#!/bin/bash
# r6 session C: the grid cut on large rotated tables - tests, the step, the bench leg (alone after the headline, and in the default run)
export EPS_TUNING_FROM_ENV=1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c
mkdir -p $O
cd $R
(timeout 1500 python -m pytest tests/test_gpu_rotated_frame.py tests/test_gpu_mfma_i8.py tests/test_gpu_full_size.py -m gpu -q -x --timeout 1400 2>&1 | tail -15 > $O/tests.txt)
(timeout 600 python scripts/lab/r6_embedding_steps.py 10000000 10 2>&1 | grep "embedding-like" > $O/embedding_steps.txt)
(timeout 900 python bench.py --configs embedding --cpu-seconds 0 --graph-rows 0 > $O/bench_embedding_only.json 2> $O/bench_embedding_only.err)
(timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err)
tail -4 $O/tests.txt; cat $O/embedding_steps.txt
python - <<'PY'
import json
for f in ("bench_embedding_only", "bench_default"):
    try:
        j = json.loads(open("gpurun_out/r6c/%s.json" % f).readline())
        e = j["configs"].get("embedding_like_10Mx768", {})
        print(f, "headline %.1f k q/s frac %.3f | embedding %.1f k q/s %.3f ms bits %s rerank %.0f ovf %s kernel %.3f" % (j["value"] / 1e3, j["roofline"]["frac"], e.get("qps", 0) / 1e3, e.get("ms_per_step", 0), e.get("operand_bits"), e.get("rerank_rows_per_query", 0), e.get("overflow_queries"), e.get("main_kernel_ms") or 0))
    except Exception as ex:
        print(f, "failed", ex)
PY
