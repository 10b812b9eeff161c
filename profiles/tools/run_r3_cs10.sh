# round 3: candidate search -- runs of consecutive reads per workgroup (0 = persistent workgroups drawing four reads at a time)
mkdir -p gpurun_out/r3
for r in 0 16 32 64 128; do
NGM_HIP_CS_READS_PER_WG=$r NGM_HIP_HOST_TIMING=1 timeout 900 python bench.py --steps 6 --no-end-to-end --no-cpu-baseline > gpurun_out/r3/bench_cs10_r$r.log 2> gpurun_out/r3/bench_cs10_r$r.err; python - <<PY
import json,re,statistics
j=json.loads([l for l in open('gpurun_out/r3/bench_cs10_r$r.log') if l.startswith('{')][0])
o=[float(m.group(1)) for m in (re.search(r'\| order ([\d.]+) \|', l) for l in open('gpurun_out/r3/bench_cs10_r$r.err')) if m]
print('reads per workgroup $r', j['value'], j['ms_per_step'], j['kernel_ms']['candidate_search'], j['kernel_ms']['all_kernels'], 'order wait ms', round(statistics.mean(o),2) if o else None)
PY
done
timeout 2400 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r3/t_cs10.log 2>&1; tail -4 gpurun_out/r3/t_cs10.log
