import os, sys, subprocess
sys.path.insert(0, "tests")
import humanlike as H, ref_files as RF
d = "/tmp/dbg_hl"; os.makedirs(d + "/refrun", exist_ok=True)
G = H.make_genome(total_bp=120_000_000, seed=7, sine_copies=100_000)
fa = d + "/ref.fa"; H.write_fasta(fa, G)
if not os.path.exists(d + "/refrun/ref.fa"): os.link(fa, d + "/refrun/ref.fa")
se = H.make_reads(G, 12000, 150, seed=11)
H.write_fastq(d + "/se.fq", se)
args = ["-q", d + "/se.fq"] + sys.argv[1:]
r = RF.run_ngm(["-r", d + "/refrun/ref.fa", "-o", d + "/refrun/o.sam", "--affine", "-t", "1", "--no-progress"] + args, cwd=d + "/refrun", timeout=3000)
c = subprocess.run(["nextgenmap_amd/ngm-hip", "-r", fa, "-o", d + "/h.sam", "--affine"] + args, capture_output=True, text=True)
def load(p):
    x = {}
    for l in open(p):
        if l[0] != "@": x.setdefault(l.split("\t", 1)[0], []).append(l.rstrip("\n").split("\t"))
    return x
a, b = load(d + "/refrun/o.sam"), load(d + "/h.sam")
for k in a:
    ka = sorted(tuple(f[:9] + f[11:]) for f in a[k]); kb = sorted(tuple(f[:9] + f[11:]) for f in b[k])
    if ka != kb:
        print("READ", k)
        for f in a[k]: print("   ref:", f[1:9], f[11:])
        for f in b[k]: print("   hip:", f[1:9], f[11:])
