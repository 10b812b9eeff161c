mkdir -p gpurun_out/r2
timeout 1200 python -m pytest tests/test_gpu_refindex.py tests/test_gpu_pipeline.py tests/test_gpu_dropin.py tests/test_gpu_cli_golden.py -q -m gpu -x 2>&1 | tail -4
NGM_HIP_HOST_TIMING=1 timeout 600 python bench.py --personality linear --no-cpu-baseline --no-end-to-end --steps 4 > gpurun_out/r2/bench_linear_t.log 2>&1; grep "host wall" gpurun_out/r2/bench_linear_t.log | tail -4; tail -1 gpurun_out/r2/bench_linear_t.log | cut -c1-200
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r2/bench_linear_t.log") if x.startswith("{")][-1]
print(json.loads(l)["setup_s"])
PY
