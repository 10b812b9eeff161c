# round 3: config 5's shape, mapper instances per GPU
mkdir -p gpurun_out/r3
for w in 4 6 8; do
timeout 1200 python bench.py --read-len 250 --corridor 80 --layout se --subs 0.12 --indel-bases 0.03 --sensitive --workers $w --steps 3 --no-end-to-end --no-cpu-baseline > gpurun_out/r3/bench_c5_w$w.log 2> gpurun_out/r3/bench_c5_w$w.err; python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/r3/bench_c5_w$w.log') if l.startswith('{')][0])
print('config5 workers $w', round(j['value']/1e6,3), round(j['ms_per_step'],1), round(j['kernel_ms']['all_kernels'],1), round(j['kernel_ms']['sw_score'],1))
PY
done
