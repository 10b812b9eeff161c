"""Tuning helper (not a test): ngm-hip end to end on the bench genome, a few worker / batch-size settings.
usage: python profiles/tools/quick_e2e.py [reads=4000000] [genome_mbp=3100]"""
import os, re, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench as B
from nextgenmap_amd import build
from nextgenmap_amd.pipeline import Reference

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
mbp = float(sys.argv[2]) if len(sys.argv) > 2 else 3100.0
wd = tempfile.mkdtemp(prefix="ngm_e2e_")
contigs = B.make_genome(int(mbp * 1e6), seed=20240601)
ref = Reference.from_contigs(contigs, device=0, kmer=13, kmer_skip=2, bin_size=2)
fa = os.path.join(wd, "ref.fa")
open(fa, "w").write(">stub\nACGT\n")
ref.write_ngm_cache(fa)
ref.close()
rows, _, _ = B.make_reads(contigs, n, seed=5, paired=True)
f1, f2 = os.path.join(wd, "a_1.fq"), os.path.join(wd, "a_2.fq")
B.write_fastq(rows, [f1, f2])
del rows, contigs
runs = [(["--workers", "2"], {}), (["--workers", "3"], {}), (["--workers", "4"], {}), (["--workers", "6"], {}), (["--workers", "4", "-o", "/dev/null"], {}),
        (["--workers", "4", "--bam"], {})]
if len(sys.argv) > 3:
    runs = [(a.split(), {}) for a in sys.argv[3:]]   # e.g. "--workers 4 --batch-size 131072"
for extra, env in runs:
    out = os.path.join(wd, "o.sam")
    t = time.time()
    r = subprocess.run([build.CLI, "-r", fa, "-1", f1, "-2", f2, "--affine", "-s", "0.5", "--no-progress"] + (extra if "-o" in extra else extra + ["-o", out]), capture_output=True, text=True,
                       env=dict(os.environ, NGM_HIP_HOST_TIMING="1", **env))
    log = r.stdout + r.stderr
    m = re.findall(r"(Mapping pass: [0-9.]+ s, [0-9]+ reads/s|Input to output: [0-9.]+ s|GPU kernels: [0-9.]+ s)", log)
    print("\n".join([l for l in log.splitlines() if "Worker time" in l or "Pool thread" in l or "process totals" in l or "stage ms" in l][-6:]))
    print(extra, "wall %.1f s" % (time.time() - t), m, "bytes", os.path.getsize(out) if os.path.exists(out) else None, flush=True)
    if r.returncode != 0:
        print(log[-800:])
try:
    os.remove("/dev/shm/ngm_e2e_o.sam")
except OSError:
    pass
