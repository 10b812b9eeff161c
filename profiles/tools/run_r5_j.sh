#!/bin/bash
# round 5, call J: the main leg, round-4 tree against this tree on ONE box, alternating; phases of the global order replay
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
show() { python - "$1" <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(l[-1]); print(sys.argv[1], 'value %.4g ms/step %.2f'%(d['value'],d['ms_per_step']), {k:round(v,2) for k,v in d['kernel_ms'].items()}, 'iso', round(d['roofline']['candidate_search']['isolated']['ms'],2))
PY
}
for rep in 1 2; do
  (cd _r04 && timeout 600 python bench.py --steps 10 --warmup 3 --no-end-to-end --no-cpu-baseline --heavy-tail-mbp 0 > ../gpurun_out/r5j_r04_$rep.log 2>/dev/null); show gpurun_out/r5j_r04_$rep.log
  timeout 600 python bench.py --steps 10 --warmup 3 --no-end-to-end --no-cpu-baseline --heavy-tail-mbp 0 > gpurun_out/r5j_new_$rep.log 2>/dev/null; show gpurun_out/r5j_new_$rep.log
done
NGM_HIP_HOST_PAIR_CHOICE=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-end-to-end --no-cpu-baseline --heavy-tail-mbp 0 > gpurun_out/r5j_new_hostchoice.log 2>/dev/null; show gpurun_out/r5j_new_hostchoice.log
timeout 600 python bench.py --steps 10 --warmup 3 --workers 3 --no-end-to-end --no-cpu-baseline --heavy-tail-mbp 0 > gpurun_out/r5j_new_w3.log 2>/dev/null; show gpurun_out/r5j_new_w3.log
NGM_HIP_CS_PHASES=1 timeout 900 python profiles/tools/heavy_leg_only.py --steps 1 --no-cpu-baseline > gpurun_out/r5j_heavy_phases.json 2> gpurun_out/r5j_heavy_phases.err
grep "exact order replay in global" gpurun_out/r5j_heavy_phases.err | tail -6 | cut -c1-500
