#!/bin/bash
# round 5, call P: bucket replay, second version (windows of whole buckets, qualifying pass, lane per candidate); fixed slices
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_humanlike.py tests/test_gpu_cli.py tests/test_gpu_configs.py -x -q > gpurun_out/r5p_tests.log 2>&1
tail -4 gpurun_out/r5p_tests.log
NGM_HIP_CS_PHASES=1 timeout 400 python profiles/tools/heavy_leg_only.py --steps 1 --no-cpu-baseline > gpurun_out/r5p_heavy_phases.json 2> gpurun_out/r5p_heavy_phases.err
grep "order replay through buckets\|exact order replay in global" gpurun_out/r5p_heavy_phases.err | tail -4 | cut -c1-600
for v in default default2 w8 ldsbig; do
unset NGM_HIP_ORDER_BUCKET_W8 NGM_HIP_ORDER_LDS_BIG
case $v in
w8) export NGM_HIP_ORDER_BUCKET_W8=1;;
ldsbig) export NGM_HIP_ORDER_LDS_BIG=1;;
esac
NGM_HIP_ORDER_TRACE=1 timeout 400 python profiles/tools/heavy_leg_only.py --steps 3 --no-cpu-baseline > gpurun_out/r5p_heavy_leg_$v.json 2> gpurun_out/r5p_heavy_leg_$v.err
echo "$v rc $?"
python - $v <<'PY'
import json,sys
try:
    d=json.load(open('gpurun_out/r5p_heavy_leg_%s.json'%sys.argv[1]))
except Exception as e:
    print(sys.argv[1],'no result',e); sys.exit(0)
for leg in ('reads_drawn_uniformly','half_of_the_reads_from_repeats'):
    x=d[leg]; print(sys.argv[1],leg,'%.3g reads/s'%x['value'],'ms/step %.0f'%x['ms_per_step'],{k:round(v,1) for k,v in x['kernel_ms'].items()}, x['gpu_kernels_fraction_of_step'])
PY
tail -3 gpurun_out/r5p_heavy_leg_$v.err | cut -c1-200
done
