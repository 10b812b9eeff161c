# round 3: shard mode, parallel index-cache loader (timing), whole CLI suite
mkdir -p gpurun_out/r3
timeout 1800 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_cli.py tests/test_gpu_refindex.py -x -q -m gpu > gpurun_out/r3/t_shard.log 2>&1; tail -12 gpurun_out/r3/t_shard.log
NGM_HIP_LOAD_TIMING=1 timeout 1500 python bench.py --steps 3 --e2e-gz-reads 0 --cpu-t1-reads 0 --no-cpu-baseline > gpurun_out/r3/bench_load.log 2> gpurun_out/r3/bench_load.err; grep "index cache" gpurun_out/r3/bench_load.err | tail -8; python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r3/bench_load.log') if l.startswith('{')][0])
e=j['end_to_end']; print({k:e[k] for k in e if k not in ('cli_log_tail','command')}); print(e['cli_log_tail'])
PY
