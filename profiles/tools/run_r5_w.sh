#!/bin/bash
# round 5, call W: bucket replay with the all-pairs count in a scalar loop; parity; workgroups per CU
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_humanlike.py -q -k "three_order or paired_end or index_files" > gpurun_out/r5w_tests.log 2>&1
tail -3 gpurun_out/r5w_tests.log | cut -c1-300
for pc in 1 2; do
NGM_HIP_ORDER_BUCKET_PER_CU=$pc NGM_HIP_CS_PHASES=1 timeout 400 python profiles/tools/heavy_leg_only.py --steps 1 --no-cpu-baseline > gpurun_out/r5w_heavy_phases_$pc.json 2> gpurun_out/r5w_heavy_phases_$pc.err
grep "order replay through buckets" gpurun_out/r5w_heavy_phases_$pc.err | tail -2 | cut -c1-700
done
for pc in 2 1; do
NGM_HIP_ORDER_BUCKET_PER_CU=$pc timeout 400 python profiles/tools/heavy_leg_only.py --steps 3 --no-cpu-baseline > gpurun_out/r5w_heavy_leg_$pc.json 2> gpurun_out/r5w_heavy_leg_$pc.err
echo "$pc rc $?"
python - $pc <<'PY'
import json,sys
try:
    d=json.load(open('gpurun_out/r5w_heavy_leg_%s.json'%sys.argv[1]))
except Exception as e:
    print(sys.argv[1],'no result',e); sys.exit(0)
for leg in ('reads_drawn_uniformly','half_of_the_reads_from_repeats'):
    x=d[leg]; print(sys.argv[1],leg,'%.3g reads/s'%x['value'],'ms/step %.0f'%x['ms_per_step'],{k:round(v,1) for k,v in x['kernel_ms'].items()}, x['gpu_kernels_fraction_of_step'])
PY
done
