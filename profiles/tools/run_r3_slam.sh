# round 3: SLAM-seq drop-in, pair pass 1 on the GPU, goldens re-checked, whole GPU suite, default bench line
mkdir -p gpurun_out/r3 gpurun_out/golden
timeout 1500 python -m pytest tests/test_gpu_dropin.py -x -q -m gpu -k "slam or bisulfite" > gpurun_out/r3/t_slam.log 2>&1; tail -25 gpurun_out/r3/t_slam.log
timeout 1200 python oracle/make_goldens.py gpurun_out/golden > gpurun_out/r3/oracle_vs_reference_kernels.log 2>&1; tail -3 gpurun_out/r3/oracle_vs_reference_kernels.log
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r3/t_all.log 2>&1; tail -8 gpurun_out/r3/t_all.log
timeout 1500 python bench.py > gpurun_out/r3/bench_default.log 2> gpurun_out/r3/bench_default.err; tail -c 6000 gpurun_out/r3/bench_default.log; tail -5 gpurun_out/r3/bench_default.err
