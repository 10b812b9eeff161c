mkdir -p gpurun_out/r2
for w in 5 4; do
NGM_HIP_BUCKET_LOG2_WORDS=$w NGM_HIP_CS_PHASES=1 timeout 900 python bench.py --no-cpu-baseline --no-end-to-end --steps 5 > gpurun_out/r2/bench_w$w.log 2>&1; echo "log2W=$w"; tail -1 gpurun_out/r2/bench_w$w.log | cut -c1-180; grep "cs fast" gpurun_out/r2/bench_w$w.log | tail -1
done
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_cli.py tests/test_gpu_cli_golden.py tests/test_gpu_refindex.py -x -q -m gpu > gpurun_out/r2/t_default.log 2>&1; tail -3 gpurun_out/r2/t_default.log
NGM_HIP_BUCKET_LOG2_WORDS=5 timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_cli.py -x -q -m gpu > gpurun_out/r2/t_w32.log 2>&1; tail -3 gpurun_out/r2/t_w32.log
timeout 600 python profiles/tools/debug_dropin_linear.py > gpurun_out/r2/debug_dropin.log 2>&1; tail -12 gpurun_out/r2/debug_dropin.log
timeout 900 python -m pytest tests/test_gpu_bam.py -q -m gpu > gpurun_out/r2/t_bam.log 2>&1; tail -5 gpurun_out/r2/t_bam.log
