#!/bin/bash
# round 5, call F: parallel order replay + speculative pass 2 + stats plumbing: probe, then the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python profiles/tools/heavy_tail_probe.py --steps 3 > gpurun_out/r5f_probe.log 2>&1
grep "pass 1b\|step \|pair selection\|order replay:" gpurun_out/r5f_probe.log | tail -8
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r5f_tests.log 2>&1
tail -15 gpurun_out/r5f_tests.log
