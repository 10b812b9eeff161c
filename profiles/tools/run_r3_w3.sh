# round 3: mapper instances per GPU after the faster candidate search
mkdir -p gpurun_out/r3
for w in 2 3 4; do
timeout 900 python bench.py --steps 6 --workers $w --no-end-to-end --no-cpu-baseline > gpurun_out/r3/bench_w$w.log 2> gpurun_out/r3/bench_w$w.err; python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/r3/bench_w$w.log') if l.startswith('{')][0])
print('workers $w', j['value'], j['ms_per_step'], j['kernel_ms']['all_kernels'])
PY
done
