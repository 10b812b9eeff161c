#!/bin/bash
# round 5, call G: order replay over the list of used slots; heavy2 table passes; parity; the 3.1 Gbp leg's timing
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
NGM_HIP_CS_PHASES=1 timeout 600 python profiles/tools/heavy_tail_probe.py --steps 3 > gpurun_out/r5g_probe.log 2>&1
grep "heavy class\|pass 1b\|pass 3\|step \|pair selection\|order replay" gpurun_out/r5g_probe.log | tail -14 | cut -c1-400
timeout 1500 python -m pytest tests/test_gpu_humanlike.py tests/test_gpu_pipeline.py tests/test_gpu_dropin.py -x -q > gpurun_out/r5g_tests.log 2>&1
tail -6 gpurun_out/r5g_tests.log
NGM_HIP_HOST_TIMING=1 NGM_HIP_CS_PHASES=1 timeout 900 python profiles/tools/heavy_leg_only.py --steps 2 --no-cpu-baseline > gpurun_out/r5g_heavy_leg_timing.json 2> gpurun_out/r5g_heavy_leg_timing.err
grep "heavy class\|pass 1b\|pass 1 \|pass 3\|pair selection:\|order replay:" gpurun_out/r5g_heavy_leg_timing.err | sed -n 30,60p | cut -c1-420
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5g_heavy_leg_timing.json'))
for leg in ('reads_drawn_uniformly','half_of_the_reads_from_repeats'):
    x=d[leg]; print(leg,'%.3g reads/s'%x['value'],'ms/step %.0f'%x['ms_per_step'],{k:round(v,1) for k,v in x['kernel_ms'].items()}, x['gpu_kernels_fraction_of_step'])
PY
