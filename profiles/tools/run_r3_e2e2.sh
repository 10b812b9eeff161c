# round 3: ngm-hip end to end with the search layout prepared at reference load and 4 MB write pieces; 2 and 3 workers
mkdir -p gpurun_out/r3
for w in 2 3; do
NGM_HIP_WORKERS=$w NGM_HIP_HOST_TIMING=1 timeout 1500 python bench.py --steps 3 --cpu-t1-reads 0 --no-cpu-baseline --e2e-gz-reads 0 > gpurun_out/r3/bench_e2e5_w$w.log 2> gpurun_out/r3/bench_e2e5_w$w.err; python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/r3/bench_e2e5_w$w.log') if l.startswith('{')][0])
e=j['end_to_end']; print('workers $w', {k:e[k] for k in e if k not in ('cli_log_tail','command','input')}); print(e['cli_log_tail'][:5])
PY
done
timeout 1200 python -m pytest tests/test_gpu_cli.py tests/test_gpu_refindex.py tests/test_boundary.py -x -q -m gpu > gpurun_out/r3/t_cli3.log 2>&1; tail -4 gpurun_out/r3/t_cli3.log
