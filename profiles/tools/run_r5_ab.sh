#!/bin/bash
# round 5, call AB: heavy reads whose table pass overflows start over with twice the parts (instead of leaving for cs_global_kernel)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_humanlike.py tests/test_gpu_configs.py -q > gpurun_out/r5ab_tests.log 2>&1
tail -3 gpurun_out/r5ab_tests.log | cut -c1-300
NGM_HIP_CS_PHASES=1 timeout 400 python profiles/tools/heavy_leg_only.py --steps 1 --no-cpu-baseline > gpurun_out/r5ab_phases.json 2> gpurun_out/r5ab_phases.err
grep "heavy class 2" gpurun_out/r5ab_phases.err | tail -4 | cut -c1-520
timeout 400 python profiles/tools/heavy_leg_only.py --steps 3 --no-cpu-baseline > gpurun_out/r5ab_leg.json 2> gpurun_out/r5ab_leg.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5ab_leg.json'))
for leg in ('reads_drawn_uniformly','half_of_the_reads_from_repeats'):
    x=d[leg]; print(leg,'%.3g reads/s'%x['value'],'ms/step %.0f'%x['ms_per_step'],{k:round(v,1) for k,v in x['kernel_ms'].items()}, x['share_of_reads']['exact_search_global_table'], x['share_of_reads']['exact_search_lds_table'], x['accuracy'])
PY
