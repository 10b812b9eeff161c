#!/usr/bin/env python3
"""Golden SAM files for the command-line parity tests: runs the REAL reference program (oracle/_ref/ngm/ngm-core, built
from /root/reference by oracle/ngm_ref.mk) on small seeded inputs and stores inputs + expected records under
tests/golden/cli/.  Runs on CPU in the build container; the GPU tests then compare `ngm-hip` with these files even where
the reference binary is absent."""
import gzip
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import ref_files as RF  # noqa: E402
import simulate as S  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "cli")


def write_fa(path, contigs):
    with open(path, "wb") as f:
        for i, g in enumerate(contigs):
            f.write(b">chr%d\n" % (i + 1))
            b = g.tobytes()
            for o in range(0, len(b), 70):
                f.write(b[o:o + 70] + b"\n")


def gz(src, dst):
    with open(src, "rb") as a, gzip.GzipFile(dst, "wb", mtime=0) as b:
        shutil.copyfileobj(a, b)


def main():
    os.makedirs(OUT, exist_ok=True)
    contigs = S.make_genome([90000, 60001], seed=91, repeat_families=6, repeat_len=300, copies=5)
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "ref.fa")
        write_fa(fa, contigs)
        gz(fa, os.path.join(OUT, "ref.fa.gz"))
        # single-end (1 200 reads, needs >= 1 000 for the sensitivity estimate) and paired-end (700 pairs)
        se = S.make_reads(contigs, 1200, 100, seed=92, sub_rate=0.02, indel_rate=0.003)
        se[7] = (se[7][0], np.full(100, ord("N"), np.uint8), se[7][2])
        fq = os.path.join(d, "se.fq")
        S.write_fastq(fq, se)
        gz(fq, os.path.join(OUT, "se.fq.gz"))
        r1, r2 = S.make_reads(contigs, 700, 100, seed=93, sub_rate=0.02, indel_rate=0.003, paired=True)
        rng = np.random.default_rng(6)
        for k in range(0, 40, 4):
            r2[k] = (r2[k][0], S.ACGT[rng.integers(0, 4, 100)], r2[k][2])
        pe = os.path.join(d, "pe.fq")
        S.write_fastq(pe, [x for pair in zip(r1, r2) for x in pair])
        gz(pe, os.path.join(OUT, "pe.fq.gz"))
        # FASTA reads of mixed length (40-100), lower case and IUPAC characters, fewer than 1 000 reads (no estimation)
        rng2 = np.random.default_rng(8)
        mixed = os.path.join(d, "mixed.fa")
        with open(mixed, "wb") as f:
            for i, (name, seq, _) in enumerate(S.make_reads(contigs, 600, 100, seed=94, sub_rate=0.03, indel_rate=0.004)):
                s2 = seq[:int(rng2.integers(40, 101))].copy()
                if i % 7 == 0:
                    s2[: len(s2) // 2] |= 0x20
                if i % 11 == 0:
                    s2[int(rng2.integers(0, len(s2)))] = ord("R")
                f.write(b">" + name.encode() + b" some comment\n" + s2.tobytes() + b"\n")
        gz(mixed, os.path.join(OUT, "mixed.fa.gz"))
        # paired-end corner cases: mates of different / very short length, all-N mates, narrow insert window, --no-unal
        o1, o2 = S.make_reads(contigs, 1100, 100, seed=95, sub_rate=0.02, indel_rate=0.003, paired=True)
        odd = os.path.join(d, "pe_odd.fq")
        with open(odd, "wb") as f:
            for i, (a, b) in enumerate(zip(o1, o2)):
                for j, (name, seq, qual) in enumerate((a, b)):
                    L = 100
                    if (i + j) % 5 == 0:
                        L = int(rng2.integers(60, 100))
                    if i % 97 == 3 and j == 1:
                        L = int(rng2.integers(5, 20))
                    s2 = seq[:L].copy()
                    if i % 131 == 7 and j == 0:
                        s2[:] = ord("N")
                    f.write(b"@" + name.encode() + b"\n" + s2.tobytes() + b"\n+\n" + qual[:L] + b"\n")
        gz(odd, os.path.join(OUT, "pe_odd.fq.gz"))
        cases = {"pe_odd": ["-p", "-q", odd, "-X", "420", "--no-unal", "-R", "0.6"], "mixed_fasta": ["-q", mixed], "se_local": ["-q", fq], "se_endtoend": ["-q", fq, "-e"], "se_top3": ["-q", fq, "-n", "3"], "pe_local": ["-p", "-q", pe]}
        for name, extra in cases.items():
            out = os.path.join(d, name + ".sam")
            r = RF.run_ngm(["-r", fa, "-o", out, "--affine", "-t", "1", "--no-progress"] + extra, cwd=d)
            assert "Done" in (r.stdout + r.stderr), r.stdout + r.stderr
            with open(out) as f, gzip.GzipFile(os.path.join(OUT, name + ".sam.gz"), "wb", mtime=0) as g:
                for line in f:
                    if not line.startswith("@PG"):
                        g.write(line.encode())
            print(name, "written")


if __name__ == "__main__":
    main()
