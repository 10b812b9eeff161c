"""Seeded "human-like" genome + read simulator: a k-mer spectrum with a heavy tail instead of the
near-Poisson one of tests/simulate.py.  Test / bench infrastructure.

What GRCh38 has and a uniform random genome lacks, each as a component here (sizes scale with the genome):
  * isochores        -- base composition drifts between 35 % and 55 % GC over 100 kb .. 1 Mb stretches;
  * SINE-like family -- ONE ~300 bp consensus (Alu: ~10^6 copies in 3.1 Gbp, ~10 % of the genome) in a handful of sub-families
                        (1-4 % from the master), every copy 0.5-15 % diverged from its sub-family (power law: most copies old),
                        either strand, some 5'-truncated, with a poly-A tail: its k-mers are the ones whose lists exceed
                        max_kfreq and the "9 901 occurrences => unused" byte of the index (src/PrefixTable.cpp:468-478);
  * LINE-like        -- a few 6 kb consensuses, copies 5'-truncated to 0.3-6 kb, 1-12 % diverged;
  * satellites       -- tandem arrays of a 171 bp monomer (and a 5 bp one), 10^3-10^4 monomers per array, 0.5-3 % between monomers:
                        ONE read collects thousands of hits in neighbouring bins;
  * microsatellites  -- (A)n, (CA)n, (GAA)n, (TTAGGG)n ... of 20-300 bp, and low-complexity AT-rich runs;
  * segmental duplications -- 5-40 kb copies of unique sequence at 0.5-2 % (pairs of candidates with near-equal scores);
  * N runs.
Reads are drawn half from the whole genome and half FROM the repeat instances (the paths that serve 0.04 % of the reads on the
uniform genome -- lists that do not fit a bucket, queue / table overflow, the exact fall-backs, the candidate-order replay with many
hits, max_cmrs -- are the common case for those)."""
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
    COMP[_a] = _b


def revcomp(seq):
    return COMP[seq[::-1]]


_AT = np.frombuffer(b"AT", dtype=np.uint8)
_CG = np.frombuffer(b"CG", dtype=np.uint8)


def _random_seq(rng, n, gc=0.5):
    """one random byte per base: bit 0 picks inside the pair (A/T or C/G), bits 1-7 against the GC threshold"""
    b = rng.integers(0, 256, n, dtype=np.uint8)
    return np.where((b >> 1) < np.uint8(round(gc * 128)), _CG[b & 1], _AT[b & 1])


def _mutate_copies(rng, cons, n, div):
    """n copies of `cons`, copy i with substitution rate div[i] (a substitution picks one of the three other bases)."""
    L = len(cons)
    out = np.broadcast_to(cons, (n, L)).copy()
    m = rng.random((n, L)) < np.asarray(div)[:, None]
    idx = np.searchsorted(ACGT, out[m])
    out[m] = ACGT[(idx + rng.integers(1, 4, idx.size)) % 4]
    return out


def _power_div(rng, n, lo, hi, alpha=1.6):
    """divergences in [lo, hi], density rising towards hi like x^(alpha-1): most copies are old"""
    return lo + (hi - lo) * rng.random(n) ** (1.0 / alpha)


class Genome:
    def __init__(self):
        self.contigs = []       # uint8 arrays of ASCII bases
        self.repeats = []       # (contig, start, length, kind)


def make_genome(total_bp=120_000_000, n_contigs=3, seed=20260929, sine_frac=0.10, sine_copies=None, line_frac=0.12,
                sat_frac=0.02, micro_per_mb=250, segdup_frac=0.03, n_runs_per_contig=3):
    """-> Genome.  Every component's share is a fraction of the bases (GRCh38: SINE 13 %, LINE 21 %, satellites ~3 %, segmental
    duplications ~5 %); sine_copies overrides the copy number of the one SINE family (default: from sine_frac at ~280 bp)."""
    rng = np.random.default_rng(seed)
    G = Genome()
    lens = np.full(n_contigs, total_bp // n_contigs, dtype=np.int64)
    lens += np.arange(n_contigs) * 1013 + 1  # unequal, odd lengths
    # --- isochores -------------------------------------------------------------------------------------------------
    for L in lens:
        g = np.empty(L, dtype=np.uint8)
        p = 0
        while p < L:
            seg = int(min(L - p, rng.integers(100_000, 1_000_000)))
            g[p:p + seg] = _random_seq(rng, seg, gc=float(rng.uniform(0.35, 0.55)))
            p += seg
        G.contigs.append(g)
    frac = lens / lens.sum()

    def place(block_rows, lengths, kind):
        """writes row i (first lengths[i] bases) at a random place of a random contig"""
        n = len(lengths)
        ci = rng.choice(n_contigs, size=n, p=frac)
        for c in range(n_contigs):
            sel = np.nonzero(ci == c)[0]
            g = G.contigs[c]
            starts = rng.integers(0, len(g) - 10_000, sel.size)
            for s, i in zip(starts, sel):
                l = int(lengths[i])
                g[s:s + l] = block_rows[i][:l]
                G.repeats.append((c, int(s), l, kind))

    # --- segmental duplications (first: later repeats land inside them too) -----------------------------------------
    n_sd = int(total_bp * segdup_frac / 20_000)
    for _ in range(n_sd):
        c = int(rng.choice(n_contigs, p=frac))
        g = G.contigs[c]
        l = int(rng.integers(5_000, 40_000))
        s = int(rng.integers(0, len(g) - l))
        src = g[s:s + l]
        cp = _mutate_copies(rng, src, 1, [float(rng.uniform(0.005, 0.02))])[0]
        if rng.random() < 0.5:
            cp = revcomp(cp)
        c2 = int(rng.choice(n_contigs, p=frac))
        g2 = G.contigs[c2]
        s2 = int(rng.integers(0, len(g2) - l))
        g2[s2:s2 + l] = cp
        G.repeats.append((c2, s2, l, "segdup"))
        G.repeats.append((c, s, l, "segdup"))
    # --- LINE-like ---------------------------------------------------------------------------------------------------
    n_line_fam = 3
    line_total = int(total_bp * line_frac)
    for f in range(n_line_fam):
        cons = _random_seq(rng, 6000, gc=0.42)
        n = int(line_total / n_line_fam / 1800)   # mean copy ~1.8 kb after truncation
        keep = np.minimum(6000, (300 + rng.exponential(1500, n)).astype(np.int64))
        div = _power_div(rng, n, 0.01, 0.12)
        # copies in blocks (memory: n x 6000 bytes)
        for lo in range(0, n, 2000):
            hi = min(n, lo + 2000)
            cp = _mutate_copies(rng, cons, hi - lo, div[lo:hi])
            rows = []
            for i in range(hi - lo):
                row = cp[i][6000 - keep[lo + i]:]   # 5' truncation: the 3' end stays
                rows.append(revcomp(row) if rng.random() < 0.5 else row)
            place(rows, keep[lo:hi], "line")
    # --- the SINE-like family ------------------------------------------------------------------------------------------
    master = _random_seq(rng, 282, gc=0.6)
    subs = [master] + [_mutate_copies(rng, master, 1, [d])[0] for d in (0.01, 0.02, 0.03, 0.04, 0.04)]
    n_sine = int(sine_copies if sine_copies is not None else total_bp * sine_frac / 300)
    sub_of = rng.choice(len(subs), size=n_sine, p=[0.3, 0.25, 0.2, 0.1, 0.1, 0.05])
    div = _power_div(rng, n_sine, 0.005, 0.15)
    for lo in range(0, n_sine, 50_000):
        hi = min(n_sine, lo + 50_000)
        rows, lengths = [], []
        for sfam in range(len(subs)):
            sel = np.nonzero(sub_of[lo:hi] == sfam)[0]
            if sel.size == 0:
                continue
            cp = _mutate_copies(rng, subs[sfam], sel.size, div[lo:hi][sel])
            tails = rng.integers(5, 40, sel.size)
            trunc = np.where(rng.random(sel.size) < 0.2, rng.integers(0, 150, sel.size), 0)
            for i in range(sel.size):
                row = np.concatenate([cp[i][trunc[i]:], np.full(tails[i], ord("A"), dtype=np.uint8)])
                rows.append(revcomp(row) if rng.random() < 0.5 else row)
                lengths.append(len(row))
        place(rows, np.array(lengths), "sine")
    # --- satellites -----------------------------------------------------------------------------------------------------
    sat_total = int(total_bp * sat_frac)
    n_arrays = max(2, sat_total // 400_000)
    for a in range(n_arrays):
        mono_len = 171 if a % 3 != 2 else 5
        mono = _random_seq(rng, mono_len, gc=0.38)
        n_mono = int(sat_total / n_arrays / mono_len)
        arr = _mutate_copies(rng, mono, n_mono, np.full(n_mono, float(rng.uniform(0.005, 0.03)))).reshape(-1)
        c = int(rng.choice(n_contigs, p=frac))
        g = G.contigs[c]
        s = int(rng.integers(0, len(g) - len(arr)))
        g[s:s + len(arr)] = arr
        G.repeats.append((c, s, len(arr), "satellite"))
    # --- microsatellites + low complexity -------------------------------------------------------------------------------
    units = [b"A", b"T", b"CA", b"GT", b"GAA", b"TTAGGG", b"AT", b"AAAT", b"CAG"]
    n_micro = int(total_bp / 1e6 * micro_per_mb)
    rows, lengths = [], []
    for _ in range(n_micro):
        u = np.frombuffer(units[int(rng.integers(0, len(units)))], dtype=np.uint8)
        l = int(rng.integers(20, 300))
        row = np.tile(u, l // len(u) + 1)[:l].copy()
        m = rng.random(l) < 0.02
        row[m] = ACGT[rng.integers(0, 4, int(m.sum()))]
        rows.append(row)
        lengths.append(l)
    for _ in range(n_micro // 4):
        l = int(rng.integers(100, 1000))
        rows.append(_random_seq(rng, l, gc=0.08))
        lengths.append(l)
    place(rows, np.array(lengths), "micro")
    # --- N runs --------------------------------------------------------------------------------------------------------
    for c in range(n_contigs):
        g = G.contigs[c]
        for _ in range(n_runs_per_contig):
            s = int(rng.integers(0, len(g) - 60_000))
            g[s:s + int(rng.integers(10, 50_000))] = ord("N")
    return G


def write_fasta(path, G, width=60):
    with open(path, "wb") as f:
        for i, g in enumerate(G.contigs):
            f.write(b">chr%d\n" % (i + 1))
            n = len(g) // width * width
            body = np.empty((n // width, width + 1), dtype=np.uint8)
            body[:, :width] = g[:n].reshape(-1, width)
            body[:, width] = 10
            f.write(body.tobytes())
            if n < len(g):
                f.write(g[n:].tobytes() + b"\n")


def _mutate_read(rng, seq, sub_rate, indel_rate, max_indel, want):
    """substitutions everywhere (not on N), then the few indel events of the read one by one; one RNG call per read, not per base"""
    seq = seq.copy()
    L = len(seq)
    r = rng.random(2 * L)
    m = (r[:L] < sub_rate) & (seq != ord("N"))
    if m.any():
        idx = np.searchsorted(ACGT, seq[m])
        seq[m] = ACGT[(idx + rng.integers(1, 4, idx.size)) % 4]
    ev = np.nonzero(r[L:] < indel_rate)[0]
    if ev.size:
        parts, p = [], 0
        for e in ev:
            if e < p:
                continue
            parts.append(seq[p:e])
            l = int(min(max_indel, rng.geometric(0.5)))
            if rng.random() < 0.5:
                p = e + l                                   # deletion from the read
            else:
                parts.append(ACGT[rng.integers(0, 4, l)])   # insertion into the read
                p = e
        parts.append(seq[p:])
        seq = np.concatenate(parts)
    return seq[:want]


def make_reads(G, n, read_len, seed=20260930, repeat_share=0.5, sub_rate=0.01, indel_rate=0.001, max_indel=5, paired=False,
               insert_mean=350, insert_sd=35, kinds=None):
    """-> list of (name, seq, qual) (single-end) or two lists (paired-end, FR).  `repeat_share` of the reads (fragments) start
    inside a repeat instance, chosen uniformly over INSTANCES of the listed kinds (default: all) -- so the satellite arrays and
    the SINE family are heavily over-represented relative to their share of the bases."""
    rng = np.random.default_rng(seed)
    lens = np.array([len(c) for c in G.contigs], dtype=np.float64)
    reps = [r for r in G.repeats if kinds is None or r[3] in kinds]
    by_kind = {}
    for r in reps:
        by_kind.setdefault(r[3], []).append(r)
    kind_names = sorted(by_kind)
    r1, r2 = [], []
    span = (insert_mean + 4 * insert_sd) if paired else read_len
    i = 0
    while len(r1) < n:
        i += 1
        if reps and rng.random() < repeat_share:
            k = kind_names[int(rng.integers(0, len(kind_names)))]   # kinds equally likely: satellites are few but large
            c, s, l, kind = by_kind[k][int(rng.integers(0, len(by_kind[k])))]
            p = int(s + rng.integers(-read_len // 2, max(1, l - read_len // 2)))
        else:
            c = int(rng.choice(len(G.contigs), p=lens / lens.sum()))
            kind = "any"
            p = int(rng.integers(0, len(G.contigs[c]) - span - 2 * max_indel))
        g = G.contigs[c]
        p = max(0, min(p, len(g) - span - 2 * max_indel - 1))
        if paired:
            ins = int(max(read_len + 10, rng.normal(insert_mean, insert_sd)))
            frag = g[p:p + ins + 2 * max_indel]
            a = _mutate_read(rng, frag[:read_len + 3 * max_indel], sub_rate, indel_rate, max_indel, read_len)
            b = _mutate_read(rng, revcomp(frag[:ins])[:read_len + 3 * max_indel], sub_rate, indel_rate, max_indel, read_len)
            if len(a) < read_len or len(b) < read_len:
                continue
            strand = "+"
            if rng.random() < 0.5:
                a, b = b, a
                strand = "-"
            name = "r%d_%d_%d_%s_%s" % (len(r1), c, p, strand, kind)
            r1.append((name + "/1", a, b"I" * len(a)))
            r2.append((name + "/2", b, b"I" * len(b)))
        else:
            s = _mutate_read(rng, g[p:p + read_len + 3 * max_indel], sub_rate, indel_rate, max_indel, read_len)
            if len(s) < read_len:
                continue
            strand = "+"
            if rng.random() < 0.5:
                s = revcomp(s)
                strand = "-"
            r1.append(("r%d_%d_%d_%s_%s" % (len(r1), c, p, strand, kind), s, b"I" * len(s)))
    return (r1, r2) if paired else r1


def sample_starts(G, n, span, seed, repeat_share=0.5, kinds=None):
    """(contig index, start) of n fragments of `span` bases, vectorised: `repeat_share` of them start inside a repeat instance (kinds
    equally likely, instances of a kind equally likely), the others anywhere.  For bench.py's heavy-tail leg."""
    rng = np.random.default_rng(seed)
    lens = np.array([len(c) for c in G.contigs], dtype=np.int64)
    ci = rng.choice(len(G.contigs), size=n, p=lens / lens.sum())
    pos = (rng.random(n) * (lens[ci] - span - 16)).astype(np.int64)
    reps = [r for r in G.repeats if kinds is None or r[3] in kinds]
    if reps and repeat_share > 0:
        names = sorted(set(r[3] for r in reps))
        tab = {k: np.array([(r[0], r[1], r[2]) for r in reps if r[3] == k], dtype=np.int64) for k in names}
        from_rep = rng.random(n) < repeat_share
        kind = rng.integers(0, len(names), n)
        for ki, k in enumerate(names):
            sel = np.nonzero(from_rep & (kind == ki))[0]
            if sel.size == 0:
                continue
            inst = tab[k][rng.integers(0, len(tab[k]), sel.size)]
            p = inst[:, 1] + (rng.random(sel.size) * np.maximum(1, inst[:, 2] - span // 2)).astype(np.int64) - span // 4
            ci[sel] = inst[:, 0]
            pos[sel] = np.clip(p, 0, lens[inst[:, 0]] - span - 16)
    return ci, pos


def write_fastq(path, reads):
    with open(path, "wb") as f:
        for name, seq, qual in reads:
            f.write(b"@" + name.encode() + b"\n" + seq.tobytes() + b"\n+\n" + qual + b"\n")
