mkdir -p gpurun_out/r2
timeout 2400 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_cli.py tests/test_gpu_cli_golden.py tests/test_gpu_dropin.py tests/test_gpu_configs.py tests/test_gpu_bam.py -q -m gpu -x 2>&1 | tail -4
NGM_HIP_HOST_CIGAR=1 timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_cli_golden.py -q -m gpu -x 2>&1 | tail -2
for pers in linear affine; do
NGM_HIP_HOST_TIMING=1 timeout 600 python bench.py --personality $pers --no-cpu-baseline --no-end-to-end --steps 5 > gpurun_out/r2/bench_${pers}_t.log 2>&1; grep "host wall" gpurun_out/r2/bench_${pers}_t.log | tail -2; tail -1 gpurun_out/r2/bench_${pers}_t.log | cut -c1-200
done
