# round 3: the sequential turn of a paired-end batch without a 64-bit remainder per pair
mkdir -p gpurun_out/r3
timeout 1800 python -m pytest tests/test_gpu_configs.py tests/test_gpu_cli.py -x -q -m gpu > gpurun_out/r3/t_turn.log 2>&1; tail -3 gpurun_out/r3/t_turn.log
for i in 1 2; do
NGM_HIP_HOST_TIMING=1 timeout 900 python bench.py --steps 8 --no-end-to-end --no-cpu-baseline > gpurun_out/r3/bench_turn_$i.log 2> gpurun_out/r3/bench_turn_$i.err; python - <<PY
import json,re,statistics
j=json.loads([l for l in open('gpurun_out/r3/bench_turn_$i.log') if l.startswith('{')][0])
ps=[[float(x) for x in m.groups()] for m in (re.search(r'pair selection ms: pass 1 ([\d.]+) \| pass 2 ([\d.]+) \| order ([\d.]+) \| pass 3\+4 ([\d.]+)', l) for l in open('gpurun_out/r3/bench_turn_$i.err')) if m]
print(j['value'], j['ms_per_step'], j['kernel_ms']['all_kernels'], 'pair selection passes ms', [round(statistics.mean(r[k] for r in ps),2) for k in range(4)])
PY
done
