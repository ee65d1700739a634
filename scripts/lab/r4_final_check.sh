#!/bin/bash
# r4: what the driver runs at round end, on the final library: smoke, the whole GPU suite, the default bench line
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out/r4final
( python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > gpurun_out/r4final/smoke.txt
cat gpurun_out/r4final/smoke.txt
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > gpurun_out/r4final/pytest.txt
tail -8 gpurun_out/r4final/pytest.txt
( time timeout 900 python bench.py ) > gpurun_out/r4final/bench.json 2> gpurun_out/r4final/bench.err
cut -c1-300 gpurun_out/r4final/bench.json; tail -4 gpurun_out/r4final/bench.err
