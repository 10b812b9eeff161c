#!/bin/bash
# round 5, call E: heavy2 with the new class limits + sweep D; the heavy-tailed bench leg at 3.1 Gbp with roofline and cpu_baseline
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
NGM_HIP_CS_PHASES=1 timeout 600 python profiles/tools/heavy_tail_probe.py --steps 2 > gpurun_out/r5e_probe_phases.log 2>&1
grep "heavy class\|pass 1b" gpurun_out/r5e_probe_phases.log | tail -4
timeout 1500 python profiles/tools/heavy_leg_only.py --steps 4 > gpurun_out/r5e_heavy_leg.json 2> gpurun_out/r5e_heavy_leg.err
tail -c 6000 gpurun_out/r5e_heavy_leg.json; tail -5 gpurun_out/r5e_heavy_leg.err
NGM_HIP_HOST_TIMING=1 timeout 900 python profiles/tools/heavy_leg_only.py --steps 2 --no-cpu-baseline > gpurun_out/r5e_heavy_leg_timing.json 2> gpurun_out/r5e_heavy_leg_timing.err
grep "pass 1b\|pass 1 \|pass 3\|pair selection:\|host wall" gpurun_out/r5e_heavy_leg_timing.err | tail -24
