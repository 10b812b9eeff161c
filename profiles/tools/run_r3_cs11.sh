# round 3: candidate search of a batch in k launches (the other instance's order replay gets onto the CUs between them)
mkdir -p gpurun_out/r3
for k in 1 2 4 8; do
NGM_HIP_CS_SPLIT=$k NGM_HIP_HOST_TIMING=1 timeout 900 python bench.py --steps 6 --no-end-to-end --no-cpu-baseline > gpurun_out/r3/bench_cs11_k$k.log 2> gpurun_out/r3/bench_cs11_k$k.err; python - <<PY
import json,re,statistics
j=json.loads([l for l in open('gpurun_out/r3/bench_cs11_k$k.log') if l.startswith('{')][0])
o=[float(m.group(1)) for m in (re.search(r'\| order ([\d.]+) \|', l) for l in open('gpurun_out/r3/bench_cs11_k$k.err')) if m]
print('launches per batch $k', j['value'], j['ms_per_step'], j['kernel_ms']['candidate_search'], j['kernel_ms']['all_kernels'], 'order wait ms', round(statistics.mean(o),2) if o else None)
PY
done
NGM_HIP_CS_SPLIT=4 timeout 1200 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r3/t_cs11.log 2>&1; tail -3 gpurun_out/r3/t_cs11.log
