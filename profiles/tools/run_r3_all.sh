# round 3: whole GPU suite on the current tree + default bench line with the pre-pass timing of ngm-hip
mkdir -p gpurun_out/r3
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r3/t_all2.log 2>&1; tail -8 gpurun_out/r3/t_all2.log
NGM_HIP_HOST_TIMING=1 timeout 1500 python bench.py --cpu-t1-reads 0 > gpurun_out/r3/bench_default3.log 2> gpurun_out/r3/bench_default3.err; python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r3/bench_default3.log') if l.startswith('{')][0])
print(j['value'], j['ms_per_step'], j['kernel_ms'])
e=j['end_to_end']; print({k:e[k] for k in e if k not in ('cli_log_tail','command')}); print(e['cli_log_tail'])
PY
