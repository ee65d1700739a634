for m in 10 8 7 6 5; do echo "mantissa $m"; EPS_MFMA_MANTISSA=$m timeout 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --graph-rows 0 --recall-queries 256 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'], d['stats']['rerank_rows_per_query'], d['recall_at_10'], d['stats']['overflow_queries'])"; done
