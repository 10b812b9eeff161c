# round 3: candidate search -- reads drawn four at a time from the launch's counter
mkdir -p gpurun_out/r3
NGM_HIP_CS_PHASES=1 timeout 900 python bench.py --steps 5 --no-end-to-end --no-cpu-baseline > gpurun_out/r3/bench_cs7.log 2> gpurun_out/r3/bench_cs7.err; python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/r3/bench_cs7.log') if l.startswith('{')][0])
print(j['value'], j['ms_per_step'], j['kernel_ms'], j['roofline']['frac'])
PY
grep "cs fast\|in front" gpurun_out/r3/bench_cs7.err | tail -2
for w in 6 5; do
NGM_HIP_CS_CANON_WPE=$w timeout 900 python bench.py --steps 5 --no-end-to-end --no-cpu-baseline > gpurun_out/r3/bench_cs7_w$w.log 2> gpurun_out/r3/bench_cs7_w$w.err; python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/r3/bench_cs7_w$w.log') if l.startswith('{')][0])
print('wpe $w', j['value'], j['ms_per_step'], j['kernel_ms']['candidate_search'])
PY
done
timeout 2400 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_configs.py tests/test_gpu_cli.py -x -q -m gpu > gpurun_out/r3/t_cs7.log 2>&1; tail -4 gpurun_out/r3/t_cs7.log
