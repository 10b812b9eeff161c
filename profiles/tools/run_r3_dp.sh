# round 3: packed DP kernels (affine score without the Eh/Ev clamp, affine align flags, linear align two pairs per lane): parity + timing
mkdir -p gpurun_out/r3
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_affine.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r3/t_dp.log 2>&1; tail -8 gpurun_out/r3/t_dp.log
timeout 900 python bench.py --no-end-to-end --no-cpu-baseline > gpurun_out/r3/bench_dp_affine.log 2> gpurun_out/r3/bench_dp_affine.err; python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r3/bench_dp_affine.log') if l.startswith('{')][0])
print(j['value'], j['ms_per_step'], j['sw_gcells_per_s'], j['kernel_ms'])
PY
timeout 900 python bench.py --no-end-to-end --no-cpu-baseline --personality linear > gpurun_out/r3/bench_dp_linear.log 2> gpurun_out/r3/bench_dp_linear.err; python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r3/bench_dp_linear.log') if l.startswith('{')][0])
print(j['value'], j['ms_per_step'], j['sw_gcells_per_s'], j['kernel_ms'])
PY
tail -3 gpurun_out/r3/bench_dp_linear.err
