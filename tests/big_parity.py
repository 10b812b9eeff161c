#!/usr/bin/env python3
"""One-off large parity run (not collected by pytest): ngm-hip vs the reference program (`ngm --affine -t 1`) on a 60 Mbp
repeat-rich synthetic genome, 200 000 single-end reads and 100 000 pairs.  Prints how many SAM records differ."""
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import ref_files as RF
import simulate as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "nextgenmap_amd", "ngm-hip")


def recs(path, pe):
    out = {}
    for line in open(path):
        if line.startswith("@"):
            continue
        f = line.rstrip("\n").split("\t")
        out.setdefault((f[0], int(f[1]) & 0xC0 if pe else 0), []).append(tuple(f[1:]))
    return {k: sorted(v) for k, v in out.items()}


def main():
    d = tempfile.mkdtemp(dir="/tmp")
    rng = np.random.default_rng(123)
    t = time.time()
    contigs = S.make_genome([25_000_000, 20_000_000, 15_000_001], seed=201, repeat_families=60, repeat_len=800, copies=12, divergence=0.01)
    fa = os.path.join(d, "ref.fa")
    S.write_fasta(fa, contigs) if False else None
    with open(fa, "wb") as f:
        for i, g in enumerate(contigs):
            f.write(b">chr%d\n" % (i + 1))
            b = g.tobytes()
            for o in range(0, len(b), 60):
                f.write(b[o:o + 60] + b"\n")
    print("genome written %.0fs" % (time.time() - t), flush=True)
    n_se = int(os.environ.get("BIG_SE", "200000"))
    only = os.environ.get("BIG_ONLY", "")
    n_pe = int(os.environ.get("BIG_PE", "100000"))
    se = S.make_reads(contigs, n_se, 125, seed=202, sub_rate=0.015, indel_rate=0.002)
    fq = os.path.join(d, "se.fq")
    S.write_fastq(fq, se)
    r1, r2 = S.make_reads(contigs, n_pe, 125, seed=203, sub_rate=0.015, indel_rate=0.002, paired=True)
    if os.environ.get("BIG_RAGGED"):  # trimmed reads: every third read loses up to 45 bases at its 3' end
        def trim(rs):
            return [(nm, sq[:len(sq) - int(rng.integers(1, 46))], ql[:0]) if i % 3 == 0 else (nm, sq, ql) for i, (nm, sq, ql) in enumerate(rs)]
        r1, r2 = trim(r1), trim(r2)
        r1 = [(nm, sq, b"I" * len(sq)) for nm, sq, ql in r1]
        r2 = [(nm, sq, b"I" * len(sq)) for nm, sq, ql in r2]
    pe = os.path.join(d, "pe.fq")
    S.write_fastq(pe, [x for pair in zip(r1, r2) for x in pair])
    print("reads written %.0fs" % (time.time() - t), flush=True)
    for name, args, is_pe in (("se", ["-q", fq], False), ("pe", ["-p", "-q", pe], True), ("topn", ["-q", fq, "-n", "4"], False),
                              ("topn-strata", ["-q", fq, "-n", "3", "--strata"], False), ("pe-strata", ["-p", "-q", pe, "--strata"], True),
                              ("pe-e2e", ["-p", "-q", pe, "-e"], True), ("pe-window", ["-p", "-q", pe, "-I", "250", "-X", "420"], True)):
        if only and name not in only.split(","):
            continue
        t = time.time()
        r = RF.run_ngm(["-r", fa, "-o", os.path.join(d, name + "_ref.sam"), "--affine", "-t", "1", "--no-progress"] + args, cwd=d, timeout=3000)
        assert "Done" in (r.stdout + r.stderr), (r.stdout + r.stderr)[-2000:]
        t_ref = time.time() - t
        t = time.time()
        env = dict(os.environ, NGM_HIP_DUMP_COUNTS=os.path.join(d, name + "_counts.bin"))
        c = subprocess.run([CLI, "-r", fa, "-o", os.path.join(d, name + "_hip.sam"), "--affine", "--skip-save"] + args, capture_output=True, text=True, env=env)
        assert c.returncode == 0, c.stderr[-2000:]
        t_hip = time.time() - t
        a, b = recs(os.path.join(d, name + "_ref.sam"), is_pe), recs(os.path.join(d, name + "_hip.sam"), is_pe)
        diff = [k for k in a if a[k] != b.get(k)]
        print("%s: %d records, %d differ; reference %.0fs (1 thread, incl. index), ngm-hip %.0fs (incl. index)" % (name, len(a), len(diff), t_ref, t_hip), flush=True)
        if is_pe and os.environ.get("BIG_EARLY"):
            cnt = np.fromfile(os.path.join(d, name + "_counts.bin"), dtype=np.uint32).astype(np.int64)
            bcs = (1800000 // 125) & ~1
            early = set()
            for b0 in range(0, len(cnt), bcs):
                cum = np.cumsum(cnt[b0:b0 + bcs])
                hit = np.nonzero((cum % 1024 == 0) & (cnt[b0:b0 + bcs] > 0))[0] + b0
                early.update(int(i) // 2 for i in hit if i % 2 == 0)
            dp = sorted({int(k[0].split("_")[0][1:]) for k in diff})
            print("   pairs whose first mate's scores end a 1024-score buffer:", len(early), "; differing pairs:", len(dp), "; of those early:", len([x for x in dp if x in early]))
            print("   ngm-hip's own count:", [l for l in c.stderr.splitlines() if "Pairs lost as NextGenMap" in l])
            print("   counts of differing pairs:", [(x, int(cnt[2 * x]), int(cnt[2 * x + 1])) for x in dp[:20]])
        for k in diff[:6]:
            x, y = a[k][0], (b.get(k) or [()])[0]
            print("  ", k, [(i, u, v) for i, (u, v) in enumerate(zip(x, y)) if u != v])


if __name__ == "__main__":
    main()
