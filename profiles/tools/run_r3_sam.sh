# round 3: SAM text assembled on the GPU -- byte identity with the host formatter, end-to-end timing, SQ counters
mkdir -p gpurun_out/r3
timeout 1200 python -m pytest tests/test_gpu_cli.py -x -q -m gpu -k "sam_assembled" > gpurun_out/r3/t_sam.log 2>&1; tail -15 gpurun_out/r3/t_sam.log
timeout 1400 python profiles/tools/quick_e2e.py 10000000 3100 "--workers 2" "--workers 3" "--workers 2 --batch-size 131072" "--workers 3 --batch-size 131072" > gpurun_out/r3/e2e_profile3.log 2>&1; tail -24 gpurun_out/r3/e2e_profile3.log
bash profiles/run_r3_counters.sh > gpurun_out/r3/counters.log 2>&1; tail -5 gpurun_out/r3/counters.log; wc -l gpurun_out/r3_pmc/summary.txt
