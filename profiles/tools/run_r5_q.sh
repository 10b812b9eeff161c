#!/bin/bash
# round 5, call Q: bucket replay with four windows in flight and the table of M; parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_humanlike.py tests/test_gpu_cli.py tests/test_gpu_configs.py -q > gpurun_out/r5q_tests.log 2>&1
tail -6 gpurun_out/r5q_tests.log | cut -c1-300
timeout 900 python profiles/tools/humanlike_t1.py --reads 300000 > gpurun_out/r5q_humanlike_t1_300k.log 2>&1
tail -3 gpurun_out/r5q_humanlike_t1_300k.log | cut -c1-300
NGM_HIP_CS_PHASES=1 timeout 400 python profiles/tools/heavy_leg_only.py --steps 1 --no-cpu-baseline > gpurun_out/r5q_heavy_phases.json 2> gpurun_out/r5q_heavy_phases.err
grep "order replay through buckets\|exact order replay in global" gpurun_out/r5q_heavy_phases.err | tail -3 | cut -c1-600
grep "heavy class" gpurun_out/r5q_heavy_phases.err | tail -3 | cut -c1-400
for v in default; do
timeout 400 python profiles/tools/heavy_leg_only.py --steps 3 --no-cpu-baseline > gpurun_out/r5q_heavy_leg_$v.json 2> gpurun_out/r5q_heavy_leg_$v.err
echo "$v rc $?"
python - $v <<'PY'
import json,sys
try:
    d=json.load(open('gpurun_out/r5q_heavy_leg_%s.json'%sys.argv[1]))
except Exception as e:
    print(sys.argv[1],'no result',e); sys.exit(0)
for leg in ('reads_drawn_uniformly','half_of_the_reads_from_repeats'):
    x=d[leg]; print(sys.argv[1],leg,'%.3g reads/s'%x['value'],'ms/step %.0f'%x['ms_per_step'],{k:round(v,1) for k,v in x['kernel_ms'].items()}, x['gpu_kernels_fraction_of_step'])
PY
done
