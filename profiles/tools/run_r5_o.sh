#!/bin/bash
# round 5, call O: where does the default order path (LDS replay with its global time line + bucket kernel) stop?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
NGM_HIP_ORDER_TRACE=1 timeout 420 python profiles/tools/heavy_leg_only.py --steps 3 --no-cpu-baseline > gpurun_out/r5o_default.json 2> gpurun_out/r5o_default.err
echo "rc $?"; grep -c "order trace" gpurun_out/r5o_default.err; tail -12 gpurun_out/r5o_default.err | cut -c1-200
