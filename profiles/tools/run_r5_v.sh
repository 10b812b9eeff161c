#!/bin/bash
# round 5, call V: ablations of the v + tau phase of the bucket replay (timing only: the ranks are wrong)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for ab in 1 9 2 4 15; do
NGM_HIP_ORDER_BUCKET_ABLATE=$ab NGM_HIP_ORDER_BUCKET_PER_CU=1 NGM_HIP_CS_PHASES=1 timeout 300 python profiles/tools/heavy_leg_only.py --steps 1 --no-cpu-baseline > gpurun_out/r5v_$ab.json 2> gpurun_out/r5v_$ab.err
echo "ablate $ab"; grep "order replay through buckets" gpurun_out/r5v_$ab.err | tail -1 | cut -c150-560
done
