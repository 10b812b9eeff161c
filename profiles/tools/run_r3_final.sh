# round 3, final tree: the committed measurements (bench lines, rocprofv3 passes, SQ counters, the -t 1 examination), then the whole GPU suite
bash profiles/run_profile.sh r03 > gpurun_out/run_profile.log 2>&1; tail -3 gpurun_out/run_profile.log
bash profiles/run_r3_counters.sh > gpurun_out/run_counters.log 2>&1; tail -3 gpurun_out/run_counters.log
# the pairs at which the reference's score buffer would have filled exactly (early top1SE): ngm-core -t 1 on the first 2 M reads
timeout 2400 python bench.py --steps 1 --warmup 1 --e2e-gz-reads 0 --cpu-t1-reads 2000000 > gpurun_out/profiles/r03_bench_t1_2M_reads.log 2> gpurun_out/profiles/r03_bench_t1_2M_reads.err; python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/profiles/r03_bench_t1_2M_reads.log') if l.startswith('{')][0])
print(j['cpu_baseline']['parity_vs_reference_sam_t1'])
PY
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/profiles/r03_pytest_gpu.log 2>&1; tail -3 gpurun_out/profiles/r03_pytest_gpu.log
