#!/bin/bash
# round 5, call AC: mapper instances / stage lock on the heavy-tailed leg now that its kernels are 0.66 of the step
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in w6 w8 w4nolock w6nolock; do
unset NGM_HIP_GPU_STAGE_LOCK
W=4
case $v in w6) W=6;; w8) W=8;; w4nolock) export NGM_HIP_GPU_STAGE_LOCK=0;; w6nolock) W=6; export NGM_HIP_GPU_STAGE_LOCK=0;; esac
timeout 400 python profiles/tools/heavy_leg_only.py --steps 3 --workers $W --no-cpu-baseline > gpurun_out/r5ac_$v.json 2> gpurun_out/r5ac_$v.err
python - $v <<'PY'
import json,sys
try:
    d=json.load(open('gpurun_out/r5ac_%s.json'%sys.argv[1]))
except Exception as e:
    print(sys.argv[1],'no result'); sys.exit(0)
for leg in ('reads_drawn_uniformly','half_of_the_reads_from_repeats'):
    x=d[leg]; print(sys.argv[1],leg,'%.3g reads/s'%x['value'],'ms/step %.0f'%x['ms_per_step'],{k:round(v,1) for k,v in x['kernel_ms'].items() if k in ('candidate_search','all_kernels','candidate_order_replay_on_its_own_stream')}, x['gpu_kernels_fraction_of_step'])
PY
done
