mkdir -p gpurun_out/r2
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -4
for w in 2 3; do timeout 600 python bench.py --no-cpu-baseline --no-end-to-end --steps 8 --workers $w > gpurun_out/r2/bench_w$w.log 2>&1; echo workers=$w; tail -1 gpurun_out/r2/bench_w$w.log | cut -c90-200; done
NGM_HIP_HOST_TIMING=1 timeout 1500 python bench.py --steps 5 > gpurun_out/r2/bench_full.log 2>&1; tail -1 gpurun_out/r2/bench_full.log | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(d['value'], d.get('end_to_end'), d.get('cpu_baseline'))"
