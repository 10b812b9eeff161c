#!/bin/bash
# round 5, call AA: why reads leave cs_heavy2_kernel for cs_global_kernel (15 % of the heavy leg's GPU time)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
NGM_HIP_CS_PHASES=1 timeout 400 python profiles/tools/heavy_leg_only.py --steps 1 --no-cpu-baseline > gpurun_out/r5aa.json 2> gpurun_out/r5aa.err
grep "sent on\|heavy class 2" gpurun_out/r5aa.err | tail -8 | cut -c1-500
grep -i "global" gpurun_out/r5aa.err | tail -4 | cut -c1-400
