"""Debug helper (not a test): the config-5 reads whose SAM record differs from the reference program's; prints the candidate
list (location, strand, votes, affine score) next to what each side reported."""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import simulate as S, ref_files as RF
import nextgenmap_amd as N
from nextgenmap_amd.pipeline import Mapper, Reference
from test_gpu_cli import _sam
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CLI = os.path.join(ROOT, "nextgenmap_amd", "ngm-hip")
d = tempfile.mkdtemp()
contigs = S.make_genome([3_000_000, 2_000_001], seed=501, repeat_families=40, repeat_len=800, copies=12, divergence=0.01)
fa = os.path.join(d, "ref.fa")
with open(fa, "wb") as f:
    for i, g in enumerate(contigs):
        f.write(b">chr%d\n" % (i + 1)); b = g.tobytes()
        for o in range(0, len(b), 60): f.write(b[o:o + 60] + b"\n")
reads = S.make_reads(contigs, 6000, 250, seed=502, sub_rate=0.12, indel_rate=0.015, max_indel=5)
fq = os.path.join(d, "reads.fq"); S.write_fastq(fq, reads)
args = ["-q", fq, "-C", "40", "--sensitive"]
r = RF.run_ngm(["-r", fa, "-o", os.path.join(d, "ref.sam"), "--affine", "-t", "1", "--no-progress"] + args, cwd=d, timeout=3000)
c = subprocess.run([CLI, "-r", fa, "-o", os.path.join(d, "hip.sam"), "--affine"] + args, capture_output=True, text=True)
sens = float(re.search(r"Estimated sensitivity: ([0-9.]+)", c.stderr).group(1))
sens = sens - 0.35 * sens
print("sensitivity used", sens)
a, b = _sam(os.path.join(d, "ref.sam")), _sam(os.path.join(d, "hip.sam"))
diff = [n for n in a if a[n] != b[n]]
print("differing:", len(diff))
names = [x[0] for x in reads]
q, cor = 252, 80
ref = Reference.from_fasta(fa, device=0)
rows = Mapper.reads_to_rows([x[1] for x in reads], q)
m = Mapper(ref, q, cor, sensitivity=sens, gap_read=33, gap_ref=33, gap_extend=3, personality=1)
offs, mx, loc, strand, votes = m.candidate_search(rows)
eng = N.Engine(q, cor, gap_read=33, gap_ref=33, gap_extend=3, personality=1)
comp = np.zeros(256, np.uint8); comp[list(b"ACGTN")] = list(b"TGCAN")
starts = [0]
for g in contigs: starts.append(0)
for nm in diff[:4]:
    i = names.index(nm)
    print("READ", nm, "\n  ref :", {k: a[nm][k] for k in ("flag", "rname", "pos", "mapq", "cigar")}, a[nm]["tags"], "\n  ours:", {k: b[nm][k] for k in ("flag", "rname", "pos", "mapq", "cigar")}, b[nm]["tags"])
    L = int(np.count_nonzero(rows[i]))
    print("  n_cand", offs[i + 1] - offs[i], "max votes", mx[i])
    for k in range(offs[i], offs[i + 1]):
        win = np.frombuffer(ref.decode(int(loc[k]) - cor // 2, ((q + cor) | 1) + 1)[1], np.uint8)[:q + cor].copy()
        qry = rows[i].copy()
        if strand[k]: qry[:L] = comp[qry[:L][::-1]]
        s = eng.BatchScore(0, win[None, :], qry[None, :])[0]
        print("    cand loc %d -> %s strand %d votes %d score %.0f" % (int(loc[k]), ref.convert(int(loc[k])), int(strand[k]), int(votes[k]), float(s)))
