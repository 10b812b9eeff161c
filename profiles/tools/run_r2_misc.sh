mkdir -p gpurun_out/r2
hipcc --offload-arch=gfx950 -O3 profiles/tools/lds_atomic_calib.hip -o gpurun_out/r2/lds_calib && gpurun_out/r2/lds_calib | tee gpurun_out/r2/lds_calib.txt; rm -f gpurun_out/r2/lds_calib
timeout 900 python -m pytest tests/test_gpu_tails.py -x -q -m gpu 2>&1 | tail -8
timeout 1500 python -m pytest tests/test_gpu_bam.py tests/test_gpu_configs.py -q -m gpu 2>&1 | tail -12
