# round 3: candidate search -- next read's characters requested after the votes; kernel time when every read leaves after phase 1 / 2 / 3
mkdir -p gpurun_out/r3
for s in 0 1 2 3; do
NGM_HIP_CS_STOP=$s NGM_HIP_CS_PHASES=1 timeout 900 python bench.py --steps 5 --no-end-to-end --no-cpu-baseline > gpurun_out/r3/bench_cs9_s$s.log 2> gpurun_out/r3/bench_cs9_s$s.err; python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/r3/bench_cs9_s$s.log') if l.startswith('{')][0])
print('stop $s', j['value'], j['ms_per_step'], j['kernel_ms']['candidate_search'])
PY
grep "cs fast\|in front" gpurun_out/r3/bench_cs9_s$s.err | tail -2
done
timeout 2400 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r3/t_cs9.log 2>&1; tail -4 gpurun_out/r3/t_cs9.log
