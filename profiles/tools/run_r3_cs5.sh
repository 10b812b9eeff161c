# round 3: candidate search -- 128-bit LDS resets, k = 13 as a constant (A/B against the generic kernel on one box)
mkdir -p gpurun_out/r3
for g in 0 1 0 1; do
if [ $g = 1 ]; then export NGM_HIP_CS_CANON_GENERIC_K=1; else unset NGM_HIP_CS_CANON_GENERIC_K; fi
NGM_HIP_CS_PHASES=1 timeout 900 python bench.py --steps 5 --no-end-to-end --no-cpu-baseline > gpurun_out/r3/bench_cs5_g$g.log 2> gpurun_out/r3/bench_cs5_g$g.err; python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/r3/bench_cs5_g$g.log') if l.startswith('{')][0])
print('generic k $g', j['value'], j['ms_per_step'], j['kernel_ms']['candidate_search'], j['roofline']['frac'])
PY
grep "cs fast" gpurun_out/r3/bench_cs5_g$g.err | tail -1
done
unset NGM_HIP_CS_CANON_GENERIC_K
timeout 2400 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r3/t_cs5.log 2>&1; tail -4 gpurun_out/r3/t_cs5.log
