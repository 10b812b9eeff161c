#!/bin/bash
# round 5, call S: where the v + tau phase of the bucket replay spends its time
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for pc in 1 3; do
NGM_HIP_ORDER_BUCKET_PER_CU=$pc NGM_HIP_CS_PHASES=1 timeout 400 python profiles/tools/heavy_leg_only.py --steps 1 --no-cpu-baseline > gpurun_out/r5s_heavy_phases_$pc.json 2> gpurun_out/r5s_heavy_phases_$pc.err
grep "order replay through buckets" gpurun_out/r5s_heavy_phases_$pc.err | tail -2 | cut -c1-700
done
