# round 3: A/B of the first batch size on ONE box (ngm-hip end to end, 10 M reads)
mkdir -p gpurun_out/r3
for d in 1 4 1 4; do
NGM_HIP_FIRST_BATCH_DIV=$d NGM_HIP_HOST_TIMING=1 timeout 1500 python bench.py --steps 1 --warmup 1 --cpu-t1-reads 0 --no-cpu-baseline --e2e-gz-reads 0 > gpurun_out/r3/bench_e2e7_d$d.log 2> gpurun_out/r3/bench_e2e7_d$d.err; python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/r3/bench_e2e7_d$d.log') if l.startswith('{')][0])
e=j['end_to_end']; print('first batch div $d', e['seconds_first_input_byte_to_sam_closed'], e['mapping_pass_s'], e['gpu_kernel_s'], e['index_load_s']); print(e['cli_log_tail'][1])
PY
done
