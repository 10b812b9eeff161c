#!/usr/bin/env python3
"""bench.py's heavy-tailed leg alone (no uniform-genome leg in front of it): the same function, the same JSON object.
  python profiles/tools/heavy_leg_only.py [--mbp 3100] [--steps 8] [--no-cpu-baseline] [--workers 4]"""
import argparse, json, os, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
ap = argparse.ArgumentParser()
ap.add_argument("--mbp", type=float, default=3100.0)
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--workers", type=int, default=4)
ap.add_argument("--reads", type=int, default=1 << 20)
ap.add_argument("--cpu-reads", type=int, default=400_000)
ap.add_argument("--no-cpu-baseline", action="store_true")
ap.add_argument("--e2e-reads", type=int, default=0, help="reads of the ngm-hip FASTQ -> SAM run on this genome (0: skip)")
ap.add_argument("--e2e-runs", type=int, default=3)
ap.add_argument("--only", choices=["both", "uniform", "repeats"], default="both")
a = ap.parse_args()
import torch
import bench as B
args = argparse.Namespace(heavy_tail_mbp=a.mbp, reads_per_step=a.reads, read_sets=4, heavy_tail_steps=a.steps, workers=2, heavy_tail_workers=a.workers,
                          no_cpu_baseline=a.no_cpu_baseline, heavy_tail_cpu_reads=a.cpu_reads, subs=0.01, indel_bases=0.0, heavy_tail_repeat_share=0.5, heavy_tail_e2e_reads=a.e2e_reads, e2e_runs=a.e2e_runs, heavy_tail_only=a.only)
torch.cuda.set_device(0)
from nextgenmap_amd.engine import load_library
load_library().ngm_host_pin_to_device_node(0)
wd = tempfile.mkdtemp(prefix="ngm_heavy_")
try:
    out = B.heavy_tail_leg(args, torch.device("cuda", 0), 0, True, True, 0.5, wd)
finally:
    shutil.rmtree(wd, ignore_errors=True)
print(json.dumps(out))
