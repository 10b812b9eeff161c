"""Seeded genome + read simulator: synthetic genome with repeat families, reads with
substitutions / indels, 50 % reverse strand, names that carry the truth (r<idx>_<contig>_<pos>_<strand>)."""
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
    COMP[a] = b


def revcomp(seq):
    return COMP[seq[::-1]]


def make_genome(contig_lens, seed=20240601, repeat_families=8, repeat_len=400, copies=6, divergence=0.02, n_runs=2):
    rng = np.random.default_rng(seed)
    contigs = []
    fams = [ACGT[rng.integers(0, 4, repeat_len)] for _ in range(repeat_families)]
    for L in contig_lens:
        g = ACGT[rng.integers(0, 4, L)].copy()
        for fam in fams:
            for _ in range(copies):
                if L <= repeat_len + 10:
                    continue
                p = int(rng.integers(0, L - repeat_len))
                c = fam.copy()
                m = rng.random(repeat_len) < divergence
                c[m] = ACGT[rng.integers(0, 4, int(m.sum()))]
                g[p:p + repeat_len] = c
        for _ in range(n_runs):
            if L > 2000:
                p = int(rng.integers(0, L - 200))
                g[p:p + int(rng.integers(5, 120))] = ord("N")
        contigs.append(g)
    return contigs


def write_fasta(path, contigs, names=None, width=70):
    with open(path, "wb") as f:
        for i, g in enumerate(contigs):
            f.write(b">" + (names[i] if names else ("chr%d" % (i + 1))).encode() if not isinstance(names, list) or isinstance(names[i], str) else names[i])
            f.write(b"\n")
            b = g.tobytes()
            for o in range(0, len(b), width):
                f.write(b[o:o + width] + b"\n")


def _mutate(rng, seq, sub_rate, indel_rate, max_indel):
    out = []
    i = 0
    while i < len(seq):
        r = rng.random()
        if r < indel_rate / 2:
            i += int(min(max_indel, rng.geometric(0.5)))
            continue
        if r < indel_rate:
            out.extend(ACGT[rng.integers(0, 4, int(min(max_indel, rng.geometric(0.5))))])
        b = seq[i]
        if rng.random() < sub_rate and b != ord("N"):
            b = ACGT[(int(np.searchsorted(ACGT, b)) + int(rng.integers(1, 4))) % 4]
        out.append(b)
        i += 1
    return np.array(out, dtype=np.uint8)


def make_reads(contigs, n, read_len, seed=20240602, sub_rate=0.01, indel_rate=0.001, max_indel=5, paired=False,
               insert_mean=350, insert_sd=35, n_rate=0.0005):
    """-> list of (name, seq, qual) for SE, or two lists for PE (FR orientation)."""
    rng = np.random.default_rng(seed)
    lens = np.array([len(c) for c in contigs], dtype=np.float64)
    r1, r2 = [], []
    for i in range(n):
        ci = int(rng.choice(len(contigs), p=lens / lens.sum()))
        g = contigs[ci]
        if paired:
            ins = int(max(read_len + 10, rng.normal(insert_mean, insert_sd)))
            if len(g) < ins + 2 * max_indel + 2:
                continue
            p = int(rng.integers(0, len(g) - ins - 2 * max_indel))
            frag = g[p:p + ins + 2 * max_indel]
            a = _mutate(rng, frag[:read_len + max_indel], sub_rate, indel_rate, max_indel)[:read_len]
            b = _mutate(rng, revcomp(frag[:ins])[:read_len + max_indel], sub_rate, indel_rate, max_indel)[:read_len]
            if rng.random() < 0.5:
                a, b = b, a
                strand = "-"
            else:
                strand = "+"
            name = "r%d_%d_%d_%s" % (i, ci, p, strand)
            r1.append((name + "/1", a, b"I" * len(a)))
            r2.append((name + "/2", b, b"I" * len(b)))
        else:
            p = int(rng.integers(0, len(g) - read_len - max_indel))
            s = _mutate(rng, g[p:p + read_len + max_indel], sub_rate, indel_rate, max_indel)[:read_len]
            strand = "+"
            if rng.random() < 0.5:
                s = revcomp(s)
                strand = "-"
            if n_rate and rng.random() < 0.02:
                s = s.copy()
                s[rng.integers(0, len(s), 1)] = ord("N")
            r1.append(("r%d_%d_%d_%s" % (i, ci, p, strand), s, b"I" * len(s)))
    return (r1, r2) if paired else r1


def write_fastq(path, reads):
    with open(path, "wb") as f:
        for name, seq, qual in reads:
            f.write(b"@" + name.encode() + b"\n" + seq.tobytes() + b"\n+\n" + qual + b"\n")
