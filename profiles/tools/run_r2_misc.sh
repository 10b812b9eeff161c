mkdir -p gpurun_out/r2
NGM_HIP_HOST_TIMING=1 timeout 1500 python bench.py > gpurun_out/r2/bench_default.log 2>&1; tail -1 gpurun_out/r2/bench_default.log | cut -c1-300; grep "host wall\|pair selection ms" gpurun_out/r2/bench_default.log | tail -4
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r2/bench_default.log") if x.startswith("{")][-1]
j=json.loads(l)
print(json.dumps({k:j[k] for k in ("value","ms_per_step","kernel_ms","end_to_end","cpu_baseline","stats_allreduce")}, indent=1)[:3500])
PY
timeout 600 python tests/debug_dropin_linear.py > gpurun_out/r2/debug_dropin.log 2>&1; tail -24 gpurun_out/r2/debug_dropin.log
