# round 3: bisulfite / SLAM-seq score tables -- goldens from the reference's own -D__ALT_SCORING__ kernels, product vs oracle, drop-in
mkdir -p gpurun_out/r3 gpurun_out/golden
timeout 1200 python oracle/make_goldens.py gpurun_out/golden > gpurun_out/r3/oracle_vs_reference_kernels.log 2>&1; tail -4 gpurun_out/r3/oracle_vs_reference_kernels.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "alt_scoring" > gpurun_out/r3/t_alt.log 2>&1; tail -12 gpurun_out/r3/t_alt.log
timeout 1500 python -m pytest tests/test_gpu_dropin.py -x -q -m gpu -k "bisulfite" > gpurun_out/r3/t_bs.log 2>&1; tail -25 gpurun_out/r3/t_bs.log
timeout 1200 python -m pytest tests/test_gpu_cli.py -x -q -m gpu -k "sam_assembled" > gpurun_out/r3/t_sam.log 2>&1; tail -8 gpurun_out/r3/t_sam.log
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r3/t_all.log 2>&1; tail -8 gpurun_out/r3/t_all.log
