#!/bin/bash
# r5: one-pass search - result counters through host-mapped words (EPS_S8_HOST_WORDS) and the two-launch form (EPS_S8_TWO_LAUNCHES: the pass quantises
# its queries, the re-rank leaves the state clean) against the r4 form (prep launch + pass + re-rank + a device-to-host copy of the counters)
mkdir -p gpurun_out
export EPS_TUNING_FROM_ENV=1
{
  for rep in 1 2; do
    echo "r4 form (EPS_S8_HOST_WORDS=0)                   $(EPS_S8_HOST_WORDS=0 python scripts/prof_single_query.py 1000000 768 2>&1 | tail -1)"
    echo "host words, three launches (EPS_S8_TWO_LAUNCHES=0) $(EPS_S8_TWO_LAUNCHES=0 python scripts/prof_single_query.py 1000000 768 2>&1 | tail -1)"
    echo "host words, two launches (default)              $(python scripts/prof_single_query.py 1000000 768 2>&1 | tail -1)"
  done
  timeout 900 python -m pytest tests/test_gpu_mfma_i8.py tests/test_gpu_full_size.py -m gpu -x -q -k "one_pass or one_to_four or configs1 or single or handful" 2>&1 | tail -5
} > gpurun_out/r5_one_pass_host_words.txt 2>&1
cat gpurun_out/r5_one_pass_host_words.txt
