#!/bin/bash
# round 5, call M: big segments by the wave; order stream priority experiment
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_humanlike.py tests/test_gpu_cli.py tests/test_gpu_configs.py -x -q > gpurun_out/r5m_tests.log 2>&1
tail -4 gpurun_out/r5m_tests.log
NGM_HIP_CS_PHASES=1 timeout 900 python profiles/tools/heavy_leg_only.py --steps 1 --no-cpu-baseline > gpurun_out/r5m_heavy_phases.json 2> gpurun_out/r5m_heavy_phases.err
grep "exact order replay in global" gpurun_out/r5m_heavy_phases.err | tail -3 | cut -c1-500
for pr in high low normal; do
NGM_HIP_ORDER_PRIORITY=$pr timeout 900 python profiles/tools/heavy_leg_only.py --steps 3 --no-cpu-baseline > gpurun_out/r5m_heavy_leg_$pr.json 2> gpurun_out/r5m_heavy_leg_$pr.err
python - $pr <<'PY'
import json,sys
d=json.load(open('gpurun_out/r5m_heavy_leg_%s.json'%sys.argv[1]))
for leg in ('reads_drawn_uniformly','half_of_the_reads_from_repeats'):
    x=d[leg]; print(sys.argv[1],leg,'%.3g reads/s'%x['value'],'ms/step %.0f'%x['ms_per_step'],{k:round(v,1) for k,v in x['kernel_ms'].items()}, x['gpu_kernels_fraction_of_step'])
PY
done
