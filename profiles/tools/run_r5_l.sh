#!/bin/bash
# round 5, call L: staged replay tail: parity + phases + heavy leg; pair walk sides
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_humanlike.py tests/test_gpu_configs.py tests/test_gpu_dropin.py -x -q -k "not slam" > gpurun_out/r5l_tests.log 2>&1
tail -5 gpurun_out/r5l_tests.log
NGM_HIP_CS_PHASES=1 timeout 900 python profiles/tools/heavy_leg_only.py --steps 1 --no-cpu-baseline > gpurun_out/r5l_heavy_phases.json 2> gpurun_out/r5l_heavy_phases.err
grep "exact order replay in global" gpurun_out/r5l_heavy_phases.err | tail -4 | cut -c1-500
timeout 900 python profiles/tools/heavy_leg_only.py --steps 3 --no-cpu-baseline > gpurun_out/r5l_heavy_leg.json 2> gpurun_out/r5l_heavy_leg.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5l_heavy_leg.json'))
for leg in ('reads_drawn_uniformly','half_of_the_reads_from_repeats'):
    x=d[leg]; print(leg,'%.3g reads/s'%x['value'],'ms/step %.0f'%x['ms_per_step'],{k:round(v,1) for k,v in x['kernel_ms'].items()}, x['gpu_kernels_fraction_of_step'])
PY
