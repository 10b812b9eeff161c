#!/bin/bash
# round 5, call Z: table passes of the largest heavy-read class (the reads that still reach cs_global_kernel: 15 % of the leg's GPU time)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for pp in 64 16; do
NGM_HIP_HEAVY_PARTS=$pp timeout 400 python profiles/tools/heavy_leg_only.py --steps 3 --no-cpu-baseline > gpurun_out/r5z_heavy_leg_$pp.json 2> gpurun_out/r5z_heavy_leg_$pp.err
echo "$pp rc $?"
python - $pp <<'PY'
import json,sys
try:
    d=json.load(open('gpurun_out/r5z_heavy_leg_%s.json'%sys.argv[1]))
except Exception as e:
    print(sys.argv[1],'no result',e); sys.exit(0)
for leg in ('reads_drawn_uniformly','half_of_the_reads_from_repeats'):
    x=d[leg]; print(sys.argv[1],leg,'%.3g reads/s'%x['value'],'ms/step %.0f'%x['ms_per_step'],{k:round(v,1) for k,v in x['kernel_ms'].items()}, x['share_of_reads']['exact_search_global_table'], x['accuracy'])
PY
done
