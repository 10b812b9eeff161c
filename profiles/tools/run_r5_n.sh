#!/bin/bash
# round 5, call N: order replay through buckets (cs_order_bucket_kernel): parity, phases, A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_humanlike.py tests/test_gpu_cli.py tests/test_gpu_configs.py -x -q > gpurun_out/r5n_tests.log 2>&1
tail -4 gpurun_out/r5n_tests.log
timeout 1500 python profiles/tools/humanlike_t1.py --reads 300000 > gpurun_out/r5n_humanlike_t1_300k.log 2>&1
tail -12 gpurun_out/r5n_humanlike_t1_300k.log | cut -c1-300
NGM_HIP_CS_PHASES=1 timeout 900 python profiles/tools/heavy_leg_only.py --steps 1 --no-cpu-baseline > gpurun_out/r5n_heavy_phases.json 2> gpurun_out/r5n_heavy_phases.err
grep "order replay through buckets\|exact order replay in global" gpurun_out/r5n_heavy_phases.err | tail -6 | cut -c1-600
for v in default w8 nobig; do
case $v in
default) export -n NGM_HIP_ORDER_BUCKET_W8 NGM_HIP_ORDER_NO_LDS_BIG;;
w8) export NGM_HIP_ORDER_BUCKET_W8=1;;
nobig) export -n NGM_HIP_ORDER_BUCKET_W8; export NGM_HIP_ORDER_NO_LDS_BIG=1;;
esac
timeout 900 python profiles/tools/heavy_leg_only.py --steps 3 --no-cpu-baseline > gpurun_out/r5n_heavy_leg_$v.json 2> gpurun_out/r5n_heavy_leg_$v.err
python - $v <<'PY'
import json,sys
d=json.load(open('gpurun_out/r5n_heavy_leg_%s.json'%sys.argv[1]))
for leg in ('reads_drawn_uniformly','half_of_the_reads_from_repeats'):
    x=d[leg]; print(sys.argv[1],leg,'%.3g reads/s'%x['value'],'ms/step %.0f'%x['ms_per_step'],{k:round(v,1) for k,v in x['kernel_ms'].items()}, x['gpu_kernels_fraction_of_step'], x['share_of_reads'])
PY
done
