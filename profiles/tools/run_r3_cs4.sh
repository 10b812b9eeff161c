# round 3: candidate search with the plane indexed by the bin's low bits (no hash multiplies)
mkdir -p gpurun_out/r3
timeout 2400 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_configs.py tests/test_gpu_cli.py -x -q -m gpu > gpurun_out/r3/t_cs4.log 2>&1; tail -4 gpurun_out/r3/t_cs4.log
NGM_HIP_CS_PHASES=1 timeout 900 python bench.py --no-end-to-end --no-cpu-baseline > gpurun_out/r3/bench_cs4.log 2> gpurun_out/r3/bench_cs4.err; python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r3/bench_cs4.log') if l.startswith('{')][0])
print(j['value'], j['ms_per_step'], j['kernel_ms'], j['roofline']['frac'])
PY
grep "cs " gpurun_out/r3/bench_cs4.err | tail -3
