import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import simulate as S
from nextgenmap_amd.pipeline import Mapper, Reference

gsize = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
nreads = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
rng = np.random.default_rng(1)
t = time.time()
ncont = 4
contigs = [S.ACGT[rng.integers(0, 4, gsize // ncont)] for _ in range(ncont)]
# repeat families: copies with 2% divergence
for f in range(200):
    fam = S.ACGT[rng.integers(0, 4, 1000)]
    for _ in range(20):
        c = contigs[rng.integers(0, ncont)]; p = rng.integers(0, len(c) - 1000)
        cp = fam.copy(); m = rng.random(1000) < 0.02; cp[m] = S.ACGT[rng.integers(0, 4, int(m.sum()))]; c[p:p+1000] = cp
print("genome gen %.1fs" % (time.time() - t)); t = time.time()
ref = Reference.from_contigs(contigs)
print("ref+index build %.1fs, entries %d, max_kfreq %d" % (time.time() - t, ref.index_entries, ref.auto_max_kfreq)); t = time.time()
# reads: vectorised sampling, 1% subs, 50% reverse
L, q, c = 150, 152, 27
ci = rng.integers(0, ncont, nreads); pos = rng.integers(0, gsize // ncont - L, nreads)
rows = np.zeros((nreads, q), np.uint8)
for k in range(ncont):
    sel = np.nonzero(ci == k)[0]
    idx = pos[sel][:, None] + np.arange(L)[None, :]
    rows[sel, :L] = contigs[k][idx]
sub = rng.random((nreads, L)) < 0.01
rows[:, :L][sub] = S.ACGT[rng.integers(0, 4, int(sub.sum()))]
rev = rng.random(nreads) < 0.5
comp = np.zeros(256, np.uint8); comp[list(b"ACGT")] = list(b"TGCA")
rows[rev, :L] = comp[rows[rev, :L][:, ::-1]]
print("reads gen %.1fs" % (time.time() - t)); t = time.time()
mp = Mapper(ref, q, c, sensitivity=0.5)
for it in range(3):
    t = time.time()
    hits, cig, md = (None, None, None)
    import ctypes as C
    from nextgenmap_amd.pipeline import HIT_DTYPE
    h = np.zeros(nreads, HIT_DTYPE); cg = np.zeros((nreads, 4 * q), np.uint8); mdd = np.zeros((nreads, 4 * q), np.uint8)
    mp.lib.ngm_mapper_map_se(mp.h, nreads, rows.ctypes.data, h.ctypes.data, cg.ctypes.data, mdd.ctypes.data)
    dt = time.time() - t
    ms = mp.last_kernel_ms()
    print("map_se %d reads: %.3fs wall -> %.2f Mreads/s; kernels ms: cs %.1f gather %.1f score %.1f select %.2f gatherA %.1f align %.1f tb %.1f (sum %.1f)" % (
        nreads, dt, nreads / dt / 1e6, *ms[:7], sum(ms[:7])))
ok = h["mapped"] == 1
truth_ok = ok & (h["contig"] == ci) & (np.abs(h["pos"].astype(np.int64) - pos) <= 13)
print("mapped %.4f correct %.4f mean cands %.2f mapq>0 %.4f" % (ok.mean(), truth_ok.mean(), h["n_candidates"].mean(), (h["mapq"] > 0).mean()))
import cProfile
