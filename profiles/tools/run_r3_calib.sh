# round 3: VALU / LDS issue-rate calibration; mapper instances per GPU for config 5 and the default workload
mkdir -p gpurun_out/r3
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/valu_rate_calib profiles/tools/valu_rate_calib.hip 2> gpurun_out/r3/valu_build.err && timeout 300 /tmp/valu_rate_calib > gpurun_out/r3/valu_rate_calib.txt 2>&1; cat gpurun_out/r3/valu_rate_calib.txt
for w in 3 4; do
timeout 1200 python bench.py --read-len 250 --corridor 80 --layout se --subs 0.12 --indel-bases 0.03 --sensitive --steps 3 --workers $w --no-end-to-end --no-cpu-baseline > gpurun_out/r3/bench_config5_w$w.log 2> gpurun_out/r3/bench_config5_w$w.err; python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/r3/bench_config5_w$w.log') if l.startswith('{')][0])
print('config5 workers $w', j['value'], j['ms_per_step'], j['kernel_ms']['all_kernels'])
PY
timeout 900 python bench.py --workers $w --no-end-to-end --no-cpu-baseline > gpurun_out/r3/bench_default_w$w.log 2> gpurun_out/r3/bench_default_w$w.err; python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/r3/bench_default_w$w.log') if l.startswith('{')][0])
print('default workers $w', j['value'], j['ms_per_step'], j['kernel_ms']['all_kernels'])
PY
done
