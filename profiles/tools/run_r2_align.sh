mkdir -p gpurun_out/r2
timeout 2400 python -m pytest tests/test_gpu_affine.py tests/test_gpu_pipeline.py tests/test_gpu_cli.py tests/test_gpu_cli_golden.py tests/test_gpu_dropin.py tests/test_gpu_configs.py -q -m gpu -x 2>&1 | tail -4
NGM_HIP_HOST_TIMING=1 timeout 600 python bench.py --no-cpu-baseline --no-end-to-end --steps 5 > gpurun_out/r2/bench_align_pk.log 2>&1; grep "host wall" gpurun_out/r2/bench_align_pk.log | tail -2; tail -1 gpurun_out/r2/bench_align_pk.log | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['kernel_ms'], d['sw_gcells_per_s'], d['roofline']['isolated']['ms'])"
