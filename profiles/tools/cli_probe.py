#!/usr/bin/env python3
"""ngm-hip on a synthetic input with the per-stage timing on (NGM_HIP_HOST_TIMING): prints the [MAIN] / [INPUT] lines of every run.
  python profiles/tools/cli_probe.py --mbp 200 --reads 2000000 -- --bam      (extra arguments after -- go to ngm-hip; several runs: separate with ::)"""
import argparse, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
argv = sys.argv[1:]
extra = []
if "--" in argv:
    i = argv.index("--"); extra = argv[i + 1:]; argv = argv[:i]
ap = argparse.ArgumentParser()
ap.add_argument("--mbp", type=float, default=200.0)
ap.add_argument("--reads", type=int, default=2000000)
ap.add_argument("--gz", action="store_true")
a = ap.parse_args(argv)
import numpy as np
import bench as B
from nextgenmap_amd import build
from nextgenmap_amd.pipeline import Reference
wd = tempfile.mkdtemp(prefix="cli_probe_")
contigs = B.make_genome(int(a.mbp * 1e6), seed=1)
ref = Reference.from_contigs(contigs, device=0)
fa = os.path.join(wd, "ref.fa")
open(fa, "w").write(">stub\nACGT\n")
ref.write_ngm_cache(fa)
ref.close()
rows, _, _ = B.make_reads(contigs, a.reads, seed=2, paired=True)
files = [os.path.join(wd, "r_1.fq"), os.path.join(wd, "r_2.fq")]
B.write_fastq(rows, files)
if a.gz:
    for f in files:
        subprocess.run(["gzip", "-1", "-k", f], check=True)
runs, cur = [], []
for x in extra + ["::"]:
    if x == "::":
        runs.append(cur); cur = []
    else:
        cur.append(x)
env = dict(os.environ, NGM_HIP_HOST_TIMING="1")
for r in runs:
    gz = "--gz-input" in r
    r = [x for x in r if x != "--gz-input"]
    out = os.path.join(wd, "out.bam" if "--bam" in r else "out.sam")
    # (a run's output is dirty page cache for seconds after it: the next run's writes would wait for that write-back -- measured with the
    # same command twice: mapping pass 0.50 s, then 0.89 s.  Remove the files and let the kernel finish before the next run starts)
    for old in ("out.bam", "out.sam"):
        if os.path.exists(os.path.join(wd, old)):
            os.remove(os.path.join(wd, old))
    os.sync()
    cmd = [build.CLI, "-r", fa, "-1", files[0] + (".gz" if gz else ""), "-2", files[1] + (".gz" if gz else ""), "-o", out, "--affine", "--no-progress"] + r
    t = time.perf_counter()
    c = subprocess.run(cmd, capture_output=True, text=True, env=env)
    print("==== ngm-hip", " ".join(r), ("(gz input)" if gz else ""), "-> rc %d, %.2f s wall, %d output bytes" % (c.returncode, time.perf_counter() - t, os.path.getsize(out) if os.path.exists(out) else -1))
    for l in c.stderr.splitlines():
        if l.startswith("[MAIN]") or l.startswith("[INPUT]") or "error" in l or "index ready" in l:
            print("   ", l[:400])
import shutil
shutil.rmtree(wd, ignore_errors=True)
