mkdir -p gpurun_out/r2
NGM_HIP_CS_WAVES=1 NGM_HIP_CS_PHASES=1 timeout 900 python bench.py --no-cpu-baseline --no-end-to-end --steps 5 > gpurun_out/r2/bench_t1.log 2>&1; tail -1 gpurun_out/r2/bench_t1.log | cut -c1-200; grep "cs fast" gpurun_out/r2/bench_t1.log | tail -2
NGM_HIP_CS_WAVES=2 NGM_HIP_CS_PHASES=1 timeout 900 python bench.py --no-cpu-baseline --no-end-to-end --steps 5 > gpurun_out/r2/bench_t2.log 2>&1; tail -1 gpurun_out/r2/bench_t2.log | cut -c1-200; grep "cs fast" gpurun_out/r2/bench_t2.log | tail -2
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_cli.py tests/test_gpu_cli_golden.py tests/test_gpu_refindex.py tests/test_gpu_dropin.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r2/t_default.log 2>&1; tail -5 gpurun_out/r2/t_default.log
NGM_HIP_BUCKET_LOG2_WORDS=5 NGM_HIP_CS_WAVES=1 timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_cli.py -x -q -m gpu > gpurun_out/r2/t_w32_t1.log 2>&1; tail -3 gpurun_out/r2/t_w32_t1.log
