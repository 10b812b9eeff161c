# the whole -m gpu suite, then the default bench and the rocprofv3 passes of round 2
mkdir -p gpurun_out/r2
timeout 3000 python -m pytest tests -q -m gpu -x > gpurun_out/r2/pytest_gpu.log 2>&1; tail -4 gpurun_out/r2/pytest_gpu.log
bash profiles/run_profile.sh > gpurun_out/r2/run_profile.log 2>&1; tail -3 gpurun_out/r2/run_profile.log
cp gpurun_out/bench_affine.log gpurun_out/r2/; cp gpurun_out/bench_linear.log gpurun_out/r2/; cp gpurun_out/bench_se.log gpurun_out/r2/
tail -1 gpurun_out/bench_affine.log | cut -c1-250
