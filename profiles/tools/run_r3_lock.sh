# round 3: GPU stage lock modes (1 = one lock, default; 0 = none; 2 = one per kind of stage), order replay on the main stream
mkdir -p gpurun_out/r3
for v in "1 0" "0 0" "2 0" "1 1" "1 0"; do set -- $v
if [ $2 = 1 ]; then export NGM_HIP_ORDER_ON_MAIN_STREAM=1; else unset NGM_HIP_ORDER_ON_MAIN_STREAM; fi
NGM_HIP_GPU_STAGE_LOCK=$1 NGM_HIP_HOST_TIMING=1 timeout 900 python bench.py --steps 8 --no-end-to-end --no-cpu-baseline > gpurun_out/r3/bench_lock.log 2> gpurun_out/r3/bench_lock.err; python - <<PY
import json,re,statistics
j=json.loads([l for l in open('gpurun_out/r3/bench_lock.log') if l.startswith('{')][0])
ps=[[float(x) for x in m.groups()] for m in (re.search(r'pair selection ms: pass 1 ([\d.]+) \| pass 2 ([\d.]+) \| order ([\d.]+) \| pass 3\+4 ([\d.]+)', l) for l in open('gpurun_out/r3/bench_lock.err')) if m]
print('lock $1 order-on-main $2:', round(j['value']/1e6,2), round(j['ms_per_step'],2), round(j['kernel_ms']['all_kernels'],2), 'cs', round(j['kernel_ms']['candidate_search'],2), 'passes', [round(statistics.mean(r[k] for r in ps),2) for k in range(4)])
PY
done
