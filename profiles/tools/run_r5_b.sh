#!/bin/bash
# round 5, call B: cs_heavy2_kernel -- probe timings, parity (humanlike + pipeline), kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=${1:-r5b}
timeout 600 python profiles/tools/heavy_tail_probe.py --steps 3 > gpurun_out/${TAG}_probe.log 2>&1
grep -v "^\[ngm-hip\] candidate order replay\|libdrm" gpurun_out/${TAG}_probe.log | tail -22
timeout 1500 python -m pytest tests/test_gpu_humanlike.py tests/test_gpu_pipeline.py -x -q -s > gpurun_out/${TAG}_tests.log 2>&1
grep -E "records,|passed|failed|error|Error" gpurun_out/${TAG}_tests.log | tail -30
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o stats -- python $R/profiles/tools/heavy_tail_probe.py --steps 2 > /tmp/prof_b.log 2>&1
cd $R
S=$(find /tmp/prof_b -name "*.db" | head -1)
python - <<PY
import subprocess, sys
src = open("profiles/summarize_rocprof.py").read().replace('HERE = os.path.dirname(os.path.abspath(__file__))', 'HERE = "gpurun_out"')
open("gpurun_out/summ.py", "w").write(src)
subprocess.run([sys.executable, "gpurun_out/summ.py", "${TAG}_heavy_probe", "$S"])
PY
head -16 gpurun_out/${TAG}_heavy_probe_kernel_stats.csv
