# round 3: canonical-bucket candidate search variants (chunk-load issue point, register budget)
mkdir -p gpurun_out/r3
for v in "1 7" "0 7" "1 8"; do set -- $v
NGM_HIP_CS_CANON_CH=$1 NGM_HIP_CS_CANON_WPE=$2 NGM_HIP_CS_PHASES=1 timeout 900 python bench.py --no-cpu-baseline --no-end-to-end --steps 5 > gpurun_out/r3/bench_canon_ch$1_w$2.log 2>&1; echo "canon ch=$1 wpe=$2"; tail -1 gpurun_out/r3/bench_canon_ch$1_w$2.log | cut -c1-200; grep "cs " gpurun_out/r3/bench_canon_ch$1_w$2.log | tail -2
done
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_cli.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r3/t_default.log 2>&1; tail -3 gpurun_out/r3/t_default.log
NGM_HIP_CBUCKET_LOG2_WORDS=6 timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_cli.py -x -q -m gpu > gpurun_out/r3/t_cw6.log 2>&1; echo "cbucket log2 words 6"; tail -3 gpurun_out/r3/t_cw6.log
