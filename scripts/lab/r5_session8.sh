#!/bin/bash
# r5 session 8: the new tests (configs[1] at full size vs the compiled reference, one-pass vs oracle, adapter range fallback)
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_mfma_i8.py tests/test_dropin.py -m gpu -x -q -k "configs1 or one_to_four or beyond_the_device or batch_entry" 2>&1 | tail -15 | cut -c1-600
