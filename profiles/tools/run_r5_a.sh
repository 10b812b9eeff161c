#!/bin/bash
# round 5, call A: the restructured paired-end selection (pair_choice_kernel + sequences) -- parity on the heavy-tailed genome, timings
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out
python profiles/tools/heavy_tail_probe.py --steps 3 > gpurun_out/r5a_probe.log 2>&1
tail -25 gpurun_out/r5a_probe.log
timeout 1200 python -m pytest tests/test_gpu_humanlike.py -x -q -k "index_files or paired_end_on" -s > gpurun_out/r5a_humanlike.log 2>&1
tail -15 gpurun_out/r5a_humanlike.log
timeout 600 python -m pytest tests/test_gpu_cli.py -x -q > gpurun_out/r5a_cli.log 2>&1
tail -5 gpurun_out/r5a_cli.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_a -o stats -- python $R/profiles/tools/heavy_tail_probe.py --steps 2 > /tmp/prof_a.log 2>&1
cd $R
S=$(find /tmp/prof_a -name "*.db" | head -1)
python - <<PY
import subprocess, sys
src = open("profiles/summarize_rocprof.py").read().replace('HERE = os.path.dirname(os.path.abspath(__file__))', 'HERE = "gpurun_out"')
open("gpurun_out/summ.py", "w").write(src)
subprocess.run([sys.executable, "gpurun_out/summ.py", "r5a_heavy_probe", "$S"])
PY
head -40 gpurun_out/r5a_heavy_probe_kernel_stats.csv
