"""Python restatement of the host-side glue of NextGenMap's score/align stages (test infrastructure):
concatenated genome layout, DecodeRefSequence, reverse complement, top1SE / MAPQ, final position.
Used together with the C oracle (oracle/) to predict what the device pipeline must output."""
import math

import numpy as np

COMP = np.arange(256, dtype=np.uint8)
for a, b in zip(b"ACGT", b"TGCA"):
    COMP[a] = b


def concat_genome(contigs):
    """SequenceProvider.cpp:289-326: 1000 N, then per contig: bases (non-ACGT -> N, upper-cased), one N if the
    length is odd, 1000 N.  Contigs of length <= 10 are skipped.  Returns (ascii array, [(start, len)], n_bases)."""
    parts = [np.full(1000, ord("N"), np.uint8)]
    geom = []
    pos = 1000
    lut = np.full(256, ord("N"), np.uint8)
    for a, b in zip(b"ACGTacgt", b"ACGTACGT"):
        lut[a] = b
    for c in contigs:
        c = np.asarray(c, dtype=np.uint8)
        if c.size <= 10:
            continue
        geom.append((pos, int(c.size)))
        parts.append(lut[c])
        pos += c.size
        if c.size & 1:
            parts.append(np.full(1, ord("N"), np.uint8))
            pos += 1
        parts.append(np.full(1000, ord("N"), np.uint8))
        pos += 1000
    g = np.concatenate(parts)
    return g, geom, int(g.size)


def decode_ref(genome, n_bases, offset, buffer_len):
    """_SequenceProvider::DecodeRefSequence (SequenceProvider.cpp:382-441) -> (ok, bytes[buffer_len])."""
    concat_len = n_bases - 1
    out = np.zeros(buffer_len, np.uint8)
    ln = buffer_len - 2
    if offset >= concat_len:
        return False, out
    end = 0
    if offset + ln > concat_len:
        end = offset + ln - concat_len
        ln -= end
    idx = 0
    if offset & 1:
        out[idx] = genome[offset]
        idx += 1
        start = offset + 1
    else:
        start = offset
    k = 2 * ((ln + 1) // 2)
    out[idx:idx + k] = genome[start:start + k]
    idx += k
    if ln & 1:
        out[idx - 1] = ord("x")
    out[idx:idx + end] = ord("x")
    return True, out


def revcomp_row(row, length):
    out = np.zeros_like(row)
    out[:length] = COMP[row[:length][::-1]]
    return out


def compute_mq(best, second):
    if best > 0 and second >= 0:
        return int(math.ceil(np.float32(60.0) * (np.float32(best) - np.float32(second)) / np.float32(best)))
    return 0


def top1(scores, keys):
    """ScoreBuffer::top1SE (ScoreBuffer.cpp:228-277) with the device pipeline's tie rule (smallest key)."""
    best = second = 0.0
    num = 0
    for s in scores:
        if s > second:
            if s > best:
                second, best, num = best, s, 1
            elif s == best:
                num += 1
                second = best
            else:
                second = s
        elif s == best:
            num += 1
    if num > 0:
        cands = [i for i, s in enumerate(scores) if s == best]
    else:
        cands = list(range(len(scores)))
    win = min(cands, key=lambda i: keys[i])
    return win, compute_mq(best, second), num, best


def convert(geom, pos):
    """_SequenceProvider::convert (SequenceProvider.cpp:111-141)."""
    starts = [s for s, _ in geom] + [geom[-1][0] + geom[-1][1] + 1000]
    import bisect
    u = bisect.bisect_right(starts, pos)
    if u == 0 or u >= len(starts):
        return None
    if starts[u] - pos < 1000:
        return None
    return u - 1, pos - starts[u - 1]
