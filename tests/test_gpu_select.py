"""-m gpu: select_top1_kernel (csrc/gather_device.h) against the reference's sequential loop.

ScoreBuffer::top1SE (src/ScoreBuffer.cpp:228-277) walks a read's scores IN ORDER keeping (best, second best, number of best ones) and
computeMQ (:34-49) turns best / second into the MAPQ.  Round 6's kernel reduces a read's candidates in parallel (a thread for reads
with up to 8 candidates, a wave for the others: on a GRCh38-like genome a read has up to ~10 000), so the test restates the
reference's loop literally and feeds both the shapes that loop is sensitive to: ties of the best score, the second best equal to the
best, zero and negative scores only, a zero in front of the first positive score, one candidate, thousands of candidates."""
import ctypes as C
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def top1_reference(scores, loc, sv):
    """ScoreBuffer::top1SE's loop, then computeMQ; the winner among equal scores is the smallest (location, strand) -- what the product
    documents for the reads whose candidate order is not replayed"""
    best, second, num = np.float32(0), np.float32(0), 0
    bi, bkey = 0, None
    for j, s in enumerate(scores):
        key = (int(loc[j]) << 1) | (int(sv[j]) & 1)
        if s > second:
            if s > best:
                second, best, num, bi, bkey = best, s, 1, j, key
            elif s == best:
                num += 1
                second = best
                if key < bkey:
                    bkey, bi = key, j
            else:
                second = s
        elif s == best:
            num += 1
            if bkey is None or key < bkey:
                bkey, bi = key, j
    if num == 0:
        keys = [(int(loc[j]) << 1) | (int(sv[j]) & 1) for j in range(len(scores))]
        bi = int(np.argmin(keys))
    mq = 0
    if best > 0 and second >= 0:
        mq = int(math.ceil(np.float32(60.0) * (np.float32(best) - np.float32(second)) / np.float32(best)))
    return bi, mq, num, float(best) if best > 0 else float(scores[bi])


def _reads(rng):
    """lists of scores with the shapes named in the module docstring"""
    out = []
    for n in (1, 2, 3, 7, 8, 9, 63, 64, 65, 200, 1357, 9383):
        for kind in range(9):
            if kind == 0:
                s = rng.integers(1, 1400, n)
            elif kind == 1:
                s = rng.integers(1, 4, n)                      # many ties
            elif kind == 2:
                s = np.full(n, 777)                            # all equal
            elif kind == 3:
                s = -rng.integers(1, 50, n)                    # negative only
            elif kind == 4:
                s = np.where(rng.random(n) < 0.5, 0, -5)       # zeros and negatives
            elif kind == 5:
                s = rng.integers(-20, 21, n)                   # around zero
            elif kind == 6:
                s = np.concatenate([[0], rng.integers(1, 900, max(n - 1, 0))])[:n]   # a zero in front
            elif kind == 7:
                s = rng.integers(1, 1400, n)
                s[rng.integers(0, n)] = s.max()                # the best twice (or once, when it hits itself)
            else:
                s = np.zeros(n, np.int64)
                s[rng.integers(0, n)] = 5                      # one positive among zeros
            out.append(np.asarray(s, np.float32))
    return out


def test_select_top1_matches_the_sequential_loop():
    from nextgenmap_amd.pipeline import _lib
    lib = _lib()
    lib.ngm_debug_select_top1.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 2 + [C.c_uint64] + [C.c_void_p] * 7
    rng = np.random.default_rng(20260930)
    lists = _reads(rng)
    # scatter the reads over several workgroups of 256, with empty reads between them and a few reads' lists stored out of order
    order = rng.permutation(len(lists))
    n_reads = 3 * len(lists) + 700
    slot = np.sort(rng.choice(n_reads, len(lists), replace=False))
    count = np.zeros(n_reads, np.uint32)
    base = np.zeros(n_reads, np.uint32)
    chunks, at = [], 0
    for k, li in enumerate(order):
        count[slot[k]] = len(lists[li])
    # storage order differs from read order (the candidate regions are compacted, but nothing requires monotone bases)
    for k in rng.permutation(len(order)):
        base[slot[k]] = at
        chunks.append(lists[order[k]])
        at += len(lists[order[k]])
    scores = np.concatenate(chunks).astype(np.float32)
    n_cand = len(scores)
    loc = rng.integers(1, 1 << 31, n_cand).astype(np.uint32)
    loc[rng.random(n_cand) < 0.1] = 12345                      # equal locations: the strand bit decides
    sv = rng.integers(0, 1 << 10, n_cand).astype(np.uint32)
    winner = np.zeros(n_reads, np.uint32)
    mapq = np.zeros(n_reads, np.int32)
    n_best = np.zeros(n_reads, np.int32)
    best = np.zeros(n_reads, np.float32)
    rc = lib.ngm_debug_select_top1(0, n_reads, base.ctypes.data, count.ctypes.data, n_cand, scores.ctypes.data, loc.ctypes.data, sv.ctypes.data,
                                   winner.ctypes.data, mapq.ctypes.data, n_best.ctypes.data, best.ctypes.data)
    assert rc == 0, lib.ngm_pipeline_last_error()
    checked = 0
    for r in range(n_reads):
        b, n = int(base[r]), int(count[r])
        if n == 0:
            assert winner[r] == 0xFFFFFFFF and mapq[r] == 0 and n_best[r] == 0 and best[r] == 0
            continue
        bi, mq, num, bs = top1_reference(scores[b:b + n], loc[b:b + n], sv[b:b + n])
        # equal keys (same location and strand twice) cannot occur in a candidate list; here they can: the first of them wins in both
        got = int(winner[r]) - b
        key = lambda j: (int(loc[b + j]) << 1) | (int(sv[b + j]) & 1)
        assert 0 <= got < n and key(got) == key(bi) and scores[b + got] == scores[b + bi], (r, n, got, bi)
        assert (int(mapq[r]), int(n_best[r]), float(best[r])) == (mq, num, bs), (r, n, scores[b:b + n][:10])
        checked += 1
    assert checked == len(lists)
