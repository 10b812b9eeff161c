# round 3: candidate search -- plane indexed by the bin's low bits, on top of the chunked draw
mkdir -p gpurun_out/r3
for w in 7 6; do
NGM_HIP_CS_CANON_WPE=$w NGM_HIP_CS_PHASES=1 timeout 900 python bench.py --steps 5 --no-end-to-end --no-cpu-baseline > gpurun_out/r3/bench_cs8_w$w.log 2> gpurun_out/r3/bench_cs8_w$w.err; python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/r3/bench_cs8_w$w.log') if l.startswith('{')][0])
print('wpe $w', j['value'], j['ms_per_step'], j['kernel_ms']['candidate_search'], j['roofline']['frac'])
PY
grep "cs fast\|in front\|cs canonical path (shape" gpurun_out/r3/bench_cs8_w$w.err | tail -3
done
timeout 2400 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r3/t_cs8.log 2>&1; tail -4 gpurun_out/r3/t_cs8.log
