#!/bin/bash
# round 5, call AD: eight mapper instances with one lock per stage kind; twelve instances
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in w8lock2 w12; do
unset NGM_HIP_GPU_STAGE_LOCK
W=8
case $v in w8lock2) export NGM_HIP_GPU_STAGE_LOCK=2;; w12) W=12;; esac
timeout 400 python profiles/tools/heavy_leg_only.py --steps 3 --workers $W --no-cpu-baseline > gpurun_out/r5ad_$v.json 2> gpurun_out/r5ad_$v.err
python - $v <<'PY'
import json,sys
try:
    d=json.load(open('gpurun_out/r5ad_%s.json'%sys.argv[1]))
except Exception as e:
    print(sys.argv[1],'no result'); sys.exit(0)
for leg in ('reads_drawn_uniformly','half_of_the_reads_from_repeats'):
    x=d[leg]; print(sys.argv[1],leg,'%.3g reads/s'%x['value'],'ms/step %.0f'%x['ms_per_step'],{k:round(v,1) for k,v in x['kernel_ms'].items() if k in ('candidate_search','all_kernels','candidate_order_replay_on_its_own_stream')}, x['gpu_kernels_fraction_of_step'])
PY
done
