#!/bin/bash
# round 5, call C: heavy2 phases; lost-pair test
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
NGM_HIP_CS_PHASES=1 timeout 600 python profiles/tools/heavy_tail_probe.py --steps 2 > gpurun_out/r5c_probe_phases.log 2>&1
grep "heavy class\|pass 1b\|step " gpurun_out/r5c_probe_phases.log | tail -12
timeout 600 python profiles/tools/heavy_tail_probe.py --steps 3 > gpurun_out/r5c_probe.log 2>&1
grep "pass 1b\|step \|pair selection" gpurun_out/r5c_probe.log | tail -6
timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_configs.py -x -q > gpurun_out/r5c_tests.log 2>&1
tail -15 gpurun_out/r5c_tests.log
