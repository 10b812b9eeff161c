import os, sys, subprocess
sys.path.insert(0, "tests")
import humanlike as H, ref_files as RF
d = "/tmp/dbg_hl"; os.makedirs(d + "/refrun", exist_ok=True)
G = H.make_genome(total_bp=120_000_000, seed=7, sine_copies=100_000)
fa = d + "/ref.fa"; H.write_fasta(fa, G)
if not os.path.exists(d + "/refrun/ref.fa"): os.link(fa, d + "/refrun/ref.fa")
r1, r2 = H.make_reads(G, 6000, 150, seed=12, paired=True)
H.write_fastq(d + "/p1.fq", r1); H.write_fastq(d + "/p2.fq", r2)
args = ["-1", d + "/p1.fq", "-2", d + "/p2.fq"] + sys.argv[1:]
r = RF.run_ngm(["-r", d + "/refrun/ref.fa", "-o", d + "/refrun/o.sam", "--affine", "-t", "1", "--no-progress"] + args, cwd=d + "/refrun", timeout=3000)
c = subprocess.run(["nextgenmap_amd/ngm-hip", "-r", fa, "-o", d + "/h.sam", "--affine"] + args, capture_output=True, text=True)
def load(p):
    x = {}
    for l in open(p):
        if l[0] != "@":
            f = l.rstrip("\n").split("\t"); x[(f[0], int(f[1]) & 0xC0)] = f
    return x
a, b = load(d + "/refrun/o.sam"), load(d + "/h.sam")
names = sorted(set(k[0] for k in a if a[k] != b.get(k)))
for nm in names[:12]:
    print("PAIR", nm)
    for m in (64, 128):
        fa_, fb_ = a[(nm, m)], b[(nm, m)]
        print("   ref:", fa_[1:9], fa_[11:])
        print("   hip:", fb_[1:9], fb_[11:])
