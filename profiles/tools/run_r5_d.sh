#!/bin/bash
# round 5, call D: heavy2 after batching the survivor loops; class boundary experiments
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
NGM_HIP_CS_PHASES=1 timeout 600 python profiles/tools/heavy_tail_probe.py --steps 2 > gpurun_out/r5d_probe_phases.log 2>&1
grep "heavy class\|pass 1b" gpurun_out/r5d_probe_phases.log | tail -5
for cls in 16384,4000000000 8192,4000000000 16384,65536 32768,131072; do
  NGM_HIP_HEAVY_CLASSES=$cls NGM_HIP_CS_PHASES=1 timeout 600 python profiles/tools/heavy_tail_probe.py --steps 2 > gpurun_out/r5d_probe_cls_$cls.log 2>&1
  echo "classes $cls"; grep "heavy class\|pass 1b\|pass 3" gpurun_out/r5d_probe_cls_$cls.log | tail -6
done
timeout 1200 python -m pytest tests/test_gpu_humanlike.py -x -q -k "index_files or paired_end_on" > gpurun_out/r5d_tests.log 2>&1
tail -3 gpurun_out/r5d_tests.log
