# round 3: page-locked buffers requested before the sensitivity estimate, quarter first batches
mkdir -p gpurun_out/r3
timeout 1800 python -m pytest tests/test_gpu_cli.py tests/test_gpu_configs.py tests/test_gpu_dropin.py -x -q -m gpu > gpurun_out/r3/t_cli4.log 2>&1; tail -4 gpurun_out/r3/t_cli4.log
NGM_HIP_HOST_TIMING=1 timeout 1500 python bench.py --steps 3 --cpu-t1-reads 0 --no-cpu-baseline --e2e-gz-reads 0 > gpurun_out/r3/bench_e2e6.log 2> gpurun_out/r3/bench_e2e6.err; python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/r3/bench_e2e6.log') if l.startswith('{')][0])
e=j['end_to_end']; print({k:e[k] for k in e if k not in ('cli_log_tail','command','input')}); print(e['cli_log_tail'][:5])
PY
