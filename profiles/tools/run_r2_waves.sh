mkdir -p gpurun_out/r2
timeout 1200 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_cli_golden.py tests/test_gpu_configs.py tests/test_gpu_tails.py -q -m gpu -x 2>&1 | tail -2
for w in 2 3; do
NGM_HIP_CS_WAVES=$w NGM_HIP_CS_PHASES=1 timeout 600 python bench.py --no-cpu-baseline --no-end-to-end --steps 5 > gpurun_out/r2/bench_waves$w.log 2>&1; echo waves=$w; grep "cs fast path" gpurun_out/r2/bench_waves$w.log | tail -3; tail -1 gpurun_out/r2/bench_waves$w.log | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['kernel_ms']['candidate_search'], d['roofline']['isolated']['ms'])"
done
