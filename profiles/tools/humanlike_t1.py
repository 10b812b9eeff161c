#!/usr/bin/env python3
"""One-off parity run on the heavy-tailed genome at scale: N paired-end reads of tests/humanlike.py's genome, half of
the fragments from repeat instances, through `ngm-core --affine -t 1` (the run ngm-hip reproduces) and `ngm-hip --affine`; every SAM line
compared.    python profiles/tools/humanlike_t1.py [--mbp 120] [--reads 2000000]"""
import argparse, os, re, subprocess, sys, tempfile, time, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench as B
import humanlike as HL
import ref_files as RF
from nextgenmap_amd import build as BLD
ap = argparse.ArgumentParser()
ap.add_argument("--mbp", type=float, default=120.0)
ap.add_argument("--reads", type=int, default=2_000_000)
ap.add_argument("--repeat-share", type=float, default=0.5)
ap.add_argument("--forced-limits", default="heavy_log2c=9,heavy_log2s=7,heavy_max0=1500,heavy_scratch=4096,gtable_pool_log2=12,order_buckets_log2=6",
                help="NGM_HIP_TEST_LIMITS of a second ngm-hip run (round 6: the paths only a GRCh38-sized index takes; tests/test_gpu_humanlike.py FORCED); empty: skip")
a = ap.parse_args()
d = tempfile.mkdtemp(prefix="ngm_t1_")
try:
    t0 = time.time()
    G = HL.make_genome(total_bp=int(a.mbp * 1e6), n_contigs=3, seed=7, sine_copies=int(100_000 * a.mbp / 120.0))
    fa = os.path.join(d, "ref.fa")
    HL.write_fasta(fa, G)
    n = a.reads & ~1
    starts = HL.sample_starts(G, n // 2, 400, seed=41, repeat_share=a.repeat_share)
    rows, _, _ = B.make_reads(G.contigs, n, seed=42, paired=True, starts=starts)
    f1, f2 = os.path.join(d, "t_1.fq"), os.path.join(d, "t_2.fq")
    B.write_fastq(rows, [f1, f2])
    print("genome %.0f Mbp, %d repeat instances; %d x 150 bp paired-end reads, %.0f %% of the fragments from repeat instances (%.0f s to make)" % (a.mbp, len(G.repeats), n, 100 * a.repeat_share, time.time() - t0), flush=True)
    KEYS = ("Done", "Candidate search", "Heavy-read kernel", "Candidate order", "Pairs lost", "Input to output", "max. k-mer frequency", "Estimated sensitivity")
    def run_hip(out, limits):
        t = time.time()
        env = dict(os.environ)
        if limits: env["NGM_HIP_TEST_LIMITS"] = limits
        c = subprocess.run([BLD.CLI, "-r", fa, "-1", f1, "-2", f2, "-o", out, "--affine"], capture_output=True, text=True, env=env)
        assert c.returncode == 0, c.stderr[-2000:]
        for line in c.stderr.splitlines():
            if any(k in line for k in KEYS):
                print("  ngm-hip%s:" % (" (forced limits)" if limits else ""), line)
        return time.time() - t
    t_hip = run_hip(os.path.join(d, "hip.sam"), "")
    t_forced = run_hip(os.path.join(d, "hip_forced.sam"), a.forced_limits) if a.forced_limits else None
    t = time.time()
    r = RF.run_ngm(["-r", fa, "-1", f1, "-2", f2, "-o", os.path.join(d, "ref.sam"), "--affine", "-t", "1", "--no-progress"], cwd=d, timeout=7200)
    t_ref = time.time() - t
    log = r.stdout + r.stderr
    assert "Done" in log, log[-1500:]
    for line in log.splitlines():
        if "Done" in line or "Estimated sensitivity" in line or "Max. k-mer frequency" in line:
            print("  ngm-core:", line.strip())
    ref = {}
    with open(os.path.join(d, "ref.sam")) as fr:
        for l in fr:
            if l[0] != "@":
                t_ = l.split("\t", 2)
                ref[(t_[0], int(t_[1]) & 0xC0)] = l
    print("ngm-core --affine -t 1: %.0f s (incl. index load); ngm-hip --affine: %.1f s (incl. index build)" % (t_ref, t_hip))
    def compare(path, what):
        same = diff = n_hip = 0
        first = []
        with open(path) as fh:
            for l in fh:
                if l[0] == "@":
                    continue
                n_hip += 1
                t_ = l.split("\t", 2)
                o = ref.get((t_[0], int(t_[1]) & 0xC0))
                if o == l:
                    same += 1
                else:
                    diff += 1
                    if len(first) < 5:
                        first.append((o, l))
        print("SAM lines%s: reference %d, ngm-hip %d; identical %d, differing %d" % (what, len(ref), n_hip, same, diff))
        for o, l in first:
            print("  reference:", (o or "<missing>").rstrip()[:300]); print("  ngm-hip  :", l.rstrip()[:300])
    compare(os.path.join(d, "hip.sam"), "")
    if t_forced is not None:
        compare(os.path.join(d, "hip_forced.sam"), " (ngm-hip under NGM_HIP_TEST_LIMITS=%s, %.1f s)" % (a.forced_limits, t_forced))
finally:
    shutil.rmtree(d, ignore_errors=True)
