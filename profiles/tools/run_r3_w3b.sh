# round 3: three / four mapper instances with 524 288 reads each per step (the launches keep their size)
mkdir -p gpurun_out/r3
for v in "2 1048576" "3 1572864" "4 2097152" "3 1572864"; do set -- $v
NGM_HIP_HOST_TIMING=1 timeout 900 python bench.py --steps 6 --workers $1 --reads-per-step $2 --no-end-to-end --no-cpu-baseline > gpurun_out/r3/bench_w3b.log 2> gpurun_out/r3/bench_w3b.err; python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/r3/bench_w3b.log') if l.startswith('{')][0])
print('workers $1 reads/step $2:', round(j['value']/1e6,2), round(j['ms_per_step'],2), round(j['kernel_ms']['all_kernels'],2))
PY
done
