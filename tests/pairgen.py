"""Seeded generator of (reference window, read) pairs shaped like the buffers NGM hands to
IAlignment::BatchScore / BatchAlign:

  ref[i] : q + c bytes  (window starting c/2 before the candidate locus; may end in 'x'/NUL filler)
  qry[i] : q bytes      (read, NUL padded; read length <= q - 1)

Kinds of pairs (mixed by `mix`):
  true    read sampled from the window at diagonal c/2 +- shift, with substitutions and indels
  decoy   unrelated random read
  edge    adversarial: N runs, NUL/'x' window tails, low-complexity repeats (ties), indels at the
          band edge, very short reads, empty reads, all-N reads
"""
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _mutate(rng, seq, sub_rate, indel_rate, max_indel):
    out = []
    i = 0
    n = len(seq)
    while i < n:
        r = rng.random()
        if r < indel_rate / 2:  # deletion from the read (ref bases skipped)
            i += int(min(max_indel, rng.geometric(0.5)))
            continue
        if r < indel_rate:  # insertion into the read
            k = int(min(max_indel, rng.geometric(0.5)))
            out.extend(ACGT[rng.integers(0, 4, k)])
        b = seq[i]
        if rng.random() < sub_rate:
            b = ACGT[(int(np.searchsorted(ACGT, b)) + int(rng.integers(1, 4))) % 4] if b in ACGT else b
        out.append(b)
        i += 1
    return np.array(out, dtype=np.uint8)


def make_pairs(n, q, c, seed=1, read_len=None, sub_rate=0.02, indel_rate=0.004,
               mix=(0.6, 0.2, 0.2), n_rate=0.002):
    """Returns (ref[n, q+c] uint8, qry[n, q] uint8)."""
    rng = np.random.default_rng(seed)
    rl = q + c
    L0 = read_len if read_len is not None else q - 2
    ref = np.zeros((n, rl), dtype=np.uint8)
    qry = np.zeros((n, q), dtype=np.uint8)
    kinds = rng.choice(3, size=n, p=np.array(mix) / np.sum(mix))
    max_indel = max(1, c // 2 - 1)
    for i in range(n):
        k = kinds[i]
        win = ACGT[rng.integers(0, 4, rl)].copy()
        if k == 0 or k == 2:
            L = L0 if rng.random() < 0.8 else int(rng.integers(max(1, L0 // 2), L0 + 1))
            shift = int(rng.integers(-min(3, c // 2), min(3, c // 2) + 1))
            start = max(0, min(rl - L - max_indel, c // 2 + shift))
            src = win[start:start + L + max_indel]
            read = _mutate(rng, src, sub_rate, indel_rate, max_indel)[:L]
        else:
            L = L0
            read = ACGT[rng.integers(0, 4, L)]
        if k == 2:
            e = int(rng.integers(0, 12))
            if e == 0:    # N run in the window
                a = int(rng.integers(0, rl - 10)); win[a:a + int(rng.integers(1, 30))] = ord('N')
            elif e == 1:  # N's in the read
                idx = rng.integers(0, len(read), max(1, len(read) // 10)); read = read.copy(); read[idx] = ord('N')
            elif e == 2:  # NUL tail in the window (align-stage decode quirk / genome end)
                t = int(rng.integers(1, 6)); win[rl - t:] = 0
            elif e == 3:  # 'x' filler then NUL
                t = int(rng.integers(2, 8)); win[rl - t] = ord('x'); win[rl - t + 1:] = 0
            elif e == 4:  # low complexity: many ties
                unit = ACGT[rng.integers(0, 4, int(rng.integers(1, 4)))]
                win = np.resize(unit, rl).copy(); read = np.resize(unit, len(read)).copy()
                if len(read) > 4: read[len(read) // 2] = ACGT[rng.integers(0, 4)]
            elif e == 5:  # large indel close to the band edge
                g = max(1, c // 2 - 1); start = c // 2; h = L0 // 2
                if rng.random() < 0.5:
                    read = np.concatenate([win[start:start + h], win[start + h + g:start + h + g + (L0 - h)]])
                else:
                    read = np.concatenate([win[start:start + h], ACGT[rng.integers(0, 4, g)], win[start + h:start + L0 - g]])[:L0]
            elif e == 6:  # very short read
                read = read[:int(rng.integers(1, 8))]
            elif e == 7:  # empty read
                read = read[:0]
            elif e == 8:  # all-N read
                read = np.full(len(read), ord('N'), dtype=np.uint8)
            elif e == 9:  # lower-case / other symbols
                read = read.copy(); read[:len(read) // 3] |= 0x20
                a = int(rng.integers(0, rl - 4)); win[a:a + 3] = ord('R')
            elif e == 10:  # N against N
                a = c // 2 + int(rng.integers(0, max(1, L0 - 12)))
                win[a:a + 6] = ord('N'); read = read.copy(); read[max(0, a - c // 2):a - c // 2 + 6] = ord('N')
            elif e == 11:  # all-N window
                win[:] = ord('N')
        if n_rate > 0 and k != 2 and rng.random() < 0.05:
            idx = rng.integers(0, len(read), 1); read = read.copy(); read[idx] = ord('N')
        read = read[:q - 1]
        ref[i] = win
        qry[i, :len(read)] = read
    return ref, qry


def make_alt_pairs(n, q, c, seed=1, read_len=None, alt=1, conv_rate=0.6):
    """Pairs for the strand-specific score tables of `--bs-mapping` (alt 1) / `--slam-seq` (alt 2): make_pairs plus a random table
    choice per pair (the `direction` byte) and reads converted the way the protocol converts them on that strand --
    bisulfite: read C -> T (direction 0) or G -> A (direction 1); SLAM-seq: read T -> C or A -> G -- so that the ALT rows of the
    tables (oclDefines.cl:94-128) and the conversion branch of computeCigarMD (SWOclCigar.cpp:507-514) carry weight.
    Returns (ref, qry, dirs uint8[n])."""
    ref, qry = make_pairs(n, q, c, seed=seed, read_len=read_len)
    rng = np.random.default_rng(seed + 7919)
    dirs = (rng.random(n) < 0.5).astype(np.uint8)
    frm = {(1, 0): b"C", (1, 1): b"G", (2, 0): b"T", (2, 1): b"A"}
    to = {(1, 0): b"T", (1, 1): b"A", (2, 0): b"C", (2, 1): b"G"}
    for d in (0, 1):
        rows = np.nonzero(dirs == d)[0]
        sub = qry[rows]
        m = (sub == frm[(alt, d)][0]) & (rng.random(sub.shape) < conv_rate)
        sub[m] = to[(alt, d)][0]
        qry[rows] = sub
    return ref, qry, dirs
