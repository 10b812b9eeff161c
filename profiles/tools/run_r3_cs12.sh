# round 3: candidate search -- two compares and a scalar branch per kept bin in the completion loop
mkdir -p gpurun_out/r3
for i in 1 2; do
NGM_HIP_CS_PHASES=1 timeout 900 python bench.py --steps 6 --no-end-to-end --no-cpu-baseline > gpurun_out/r3/bench_cs12_$i.log 2> gpurun_out/r3/bench_cs12_$i.err; python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/r3/bench_cs12_$i.log') if l.startswith('{')][0])
print(j['value'], j['ms_per_step'], j['kernel_ms']['candidate_search'], j['roofline']['frac'])
PY
grep "cs fast" gpurun_out/r3/bench_cs12_$i.err | tail -1
done
timeout 2400 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_configs.py tests/test_gpu_cli.py -x -q -m gpu > gpurun_out/r3/t_cs12.log 2>&1; tail -3 gpurun_out/r3/t_cs12.log
