#!/usr/bin/env python3
"""One batch of reads against the GRCh38-like genome of bench.py's heavy-tail leg, with the library's per-pass timing on
(NGM_HIP_HOST_TIMING; NGM_HIP_CS_PHASES=1 adds the phases inside the kernels).
  python profiles/tools/heavy_tail_probe.py [--mbp 1000] [--reads 262144] [--repeat-share 0.5] [--steps 2]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
ap = argparse.ArgumentParser()
ap.add_argument("--mbp", type=float, default=1000.0)
ap.add_argument("--reads", type=int, default=262144)
ap.add_argument("--repeat-share", type=float, default=0.5)
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--se", action="store_true")
a = ap.parse_args()
os.environ["NGM_HIP_HOST_TIMING"] = "1"
import torch
import bench as B
import humanlike as HL
from nextgenmap_amd.pipeline import Mapper, Reference
G = HL.make_genome(total_bp=int(a.mbp * 1e6), n_contigs=24, seed=20260929)
ref = Reference.from_contigs(G.contigs, device=0)
print("max_kfreq", ref.auto_max_kfreq, "entries", ref.index_entries, flush=True)
paired = not a.se
starts = HL.sample_starts(G, a.reads // 2 if paired else a.reads, 400 if paired else 150, seed=20260930, repeat_share=a.repeat_share)
rows, tc, tp = B.make_reads(G.contigs, a.reads, seed=20260931, paired=paired, starts=starts)
d_rows = torch.from_numpy(rows).cuda()
m = Mapper(ref, B.Q, B.C, sensitivity=0.5, gap_read=33, gap_ref=33, gap_extend=3, personality=1)
for s in range(a.steps):
    t = time.perf_counter()
    hits, _, _ = (m.map_pe_raw if paired else m.map_se_raw)(rows, d_rows)
    print("step %d: %.1f ms, kernels %s" % (s, 1e3 * (time.perf_counter() - t), ["%.2f" % x for x in m.last_kernel_ms()]), flush=True)
print("path counters", m.path_counters(), "cs counters", m.cs_counters())
nc = hits["n_candidates"]
print("candidates per read percentiles 50/90/99/99.9/max:", np.percentile(nc, [50, 90, 99, 99.9, 100]).astype(int), "mean %.1f" % nc.mean())
print("heavy counters", m.heavy_counters())
