# round 3: where the time before the mapping pass goes; candidate search with the half chunk step
mkdir -p gpurun_out/r3
NGM_HIP_HOST_TIMING=1 NGM_HIP_LOAD_TIMING=1 NGM_HIP_CS_PHASES=1 timeout 1500 python bench.py --cpu-t1-reads 0 --no-cpu-baseline --e2e-gz-reads 0 > gpurun_out/r3/bench_e2e4.log 2> gpurun_out/r3/bench_e2e4.err; python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r3/bench_e2e4.log') if l.startswith('{')][0])
print(j['value'], j['ms_per_step'], j['kernel_ms'])
e=j['end_to_end']; print({k:e[k] for k in e if k not in ('cli_log_tail','command')}); print(e['cli_log_tail'])
PY
grep "cs " gpurun_out/r3/bench_e2e4.err | tail -3
timeout 1200 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r3/t_cs3.log 2>&1; tail -4 gpurun_out/r3/t_cs3.log
