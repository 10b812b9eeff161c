"""Host logic, no GPU: what pass 3 of the pair selection does per tied pair (csrc/mapper.cpp sort_like_reference + walk_pair, through the
host-only C-ABI entry ngm_debug_pair_walk) against a restatement of what the reference does there.

ScoreBuffer::top1PE (src/ScoreBuffer.cpp:368-413) sorts both mates' candidate lists with std::sort(sortLocationScore) -- a comparison by
score only, on the lists in CollectResultsStd's order -- and walks the candidates at or above best * pair_score_cutoff in a double loop
(CheckPairs, :463-502).  std::sort is libstdc++'s (GCC 11 here and in the reference's build; bits/stl_algo.h): introsort with a
median-of-three pivot moved to the front, an unguarded Hoare partition, heap sort once the depth limit 2 * floor(log2 n) is spent, ranges
of at most 16 left to one final insertion sort.  It is NOT stable, so which of several equally scoring candidates comes first depends on
the algorithm itself; ngm-hip therefore runs the very same std::sort on the same initial sequence.  Round 6 changed how that initial
sequence is made (a radix sort of the candidates' ranks instead of a comparison sort) and how the in-window combinations are found (one
sweep over both lists sorted by location + a bit per candidate instead of a binary search per candidate); this test pins both against
the published algorithm restated below and the plain double loop."""
import ctypes as C
import math
import zlib

import numpy as np
import pytest

UNKNOWN = 0xFFFFFFFF


# ---- libstdc++ std::sort, restated (bits/stl_algo.h of GCC 11: __sort, __introsort_loop, __unguarded_partition_pivot,
# __move_median_to_first, __unguarded_partition, __final_insertion_sort, __insertion_sort, __unguarded_linear_insert; bits/stl_heap.h:
# __make_heap, __adjust_heap, __push_heap, __pop_heap, __sort_heap) ------------------------------------------------------------------
class StdSort:
    def __init__(self, comp):
        self.comp = comp
        self.heap_sorted = 0   # ranges that ran into the depth limit

    def sort(self, v):
        n = len(v)
        if n == 0:
            return v
        self._introsort_loop(v, 0, n, 2 * (n.bit_length() - 1))
        self._final_insertion_sort(v, 0, n)
        return v

    def _introsort_loop(self, v, first, last, depth):
        while last - first > 16:
            if depth == 0:
                self._heap_sort(v, first, last)
                self.heap_sorted += 1
                return
            depth -= 1
            cut = self._partition_pivot(v, first, last)
            self._introsort_loop(v, cut, last, depth)
            last = cut

    def _partition_pivot(self, v, first, last):
        comp = self.comp
        mid = first + (last - first) // 2
        a, b, c, result = first + 1, mid, last - 1, first
        if comp(v[a], v[b]):
            if comp(v[b], v[c]):
                pick = b
            elif comp(v[a], v[c]):
                pick = c
            else:
                pick = a
        elif comp(v[a], v[c]):
            pick = a
        elif comp(v[b], v[c]):
            pick = c
        else:
            pick = b
        v[result], v[pick] = v[pick], v[result]
        lo, hi, pivot = first + 1, last, first
        while True:
            while comp(v[lo], v[pivot]):
                lo += 1
            hi -= 1
            while comp(v[pivot], v[hi]):
                hi -= 1
            if not lo < hi:
                return lo
            v[lo], v[hi] = v[hi], v[lo]
            lo += 1

    def _unguarded_linear_insert(self, v, last):
        val = v[last]
        nxt = last - 1
        while self.comp(val, v[nxt]):
            v[last] = v[nxt]
            last = nxt
            nxt -= 1
        v[last] = val

    def _insertion_sort(self, v, first, last):
        for i in range(first + 1, last):
            if self.comp(v[i], v[first]):
                val = v[i]
                v[first + 1:i + 1] = v[first:i]
                v[first] = val
            else:
                self._unguarded_linear_insert(v, i)

    def _final_insertion_sort(self, v, first, last):
        if last - first > 16:
            self._insertion_sort(v, first, first + 16)
            for i in range(first + 16, last):
                self._unguarded_linear_insert(v, i)
        else:
            self._insertion_sort(v, first, last)

    def _push_heap(self, v, first, hole, top, value):
        parent = (hole - 1) // 2
        while hole > top and self.comp(v[first + parent], value):
            v[first + hole] = v[first + parent]
            hole = parent
            parent = (hole - 1) // 2
        v[first + hole] = value

    def _adjust_heap(self, v, first, hole, length, value):
        top = hole
        child = hole
        while child < (length - 1) // 2:
            child = 2 * (child + 1)
            if self.comp(v[first + child], v[first + child - 1]):
                child -= 1
            v[first + hole] = v[first + child]
            hole = child
        if (length & 1) == 0 and child == (length - 2) // 2:
            child = 2 * (child + 1)
            v[first + hole] = v[first + child - 1]
            hole = child - 1
        self._push_heap(v, first, hole, top, value)

    def _heap_sort(self, v, first, last):   # __partial_sort(first, last, last): __heap_select (= __make_heap here) + __sort_heap
        length = last - first
        if length >= 2:
            parent = (length - 2) // 2
            while True:
                self._adjust_heap(v, first, parent, length, v[first + parent])
                if parent == 0:
                    break
                parent -= 1
        while last - first > 1:
            last -= 1
            value = v[last]
            v[last] = v[first]
            self._adjust_heap(v, first, 0, last - first, value)


def reference_order(idx, loc, sv, score, rank):
    """the list as top1PE leaves it: CollectResultsStd's order (by rank; here ranks may repeat: then by place), std::sort by score.  Without
    ranks: the total order ngm-hip documents for that case (score, location, strand)"""
    place = lambda x: (int(loc[x]), int(sv[x]) & 1)
    if rank is None or any(int(rank[x]) == UNKNOWN for x in idx):
        return sorted(idx, key=lambda x: (-float(score[x]),) + place(x)), 0
    v = sorted(idx, key=lambda x: (int(rank[x]),) + place(x))
    s = StdSort(lambda a, b: score[a] > score[b])
    return s.sort(v), s.heap_sorted


def mapq(v, score):   # computeMQ(MappedRead *), src/ScoreBuffer.cpp:42-49
    if len(v) <= 1:
        return 60
    best, second = np.float32(score[v[0]]), np.float32(score[v[1]])
    if best > 0 and second >= 0:
        return int(math.ceil(np.float32(np.float32(60.0) * (best - second)) / best))
    return 0


def expected_walk(A, B, len_a, len_b, loc, score, cutoff, min_d, max_d):
    cut = np.float32(cutoff)
    min_a, min_b = np.float32(score[A[0]]) * cut, np.float32(score[B[0]]) * cut
    na = 1
    while na < len(A) and min_a <= score[A[na]]:
        na += 1
    nb = 1
    while nb < len(B) and min_b <= score[B[nb]]:
        nb += 1
    la = loc[np.asarray(A[:na])].astype(np.int64)[:, None]
    lb = loc[np.asarray(B[:nb])].astype(np.int64)[None, :]
    cur = np.where(lb > la, lb - la + len_b, la - lb + len_a)   # ScoreBuffer.cpp:467-473: a 64-bit difference ...
    cur = (cur + 2 ** 31) % 2 ** 32 - 2 ** 31                   # ... stored in an int
    hi = max_d if max_d > 0 else 2 ** 31 - 1
    ii, jj = np.nonzero((cur > min_d) & (cur < hi))             # row-major: i major, j ascending -- the double loop's order
    ps = (score[np.asarray(A[:na])][ii] + score[np.asarray(B[:nb])][jj]).astype(np.float32)
    return ps, cur[ii, jj].astype(np.int64), np.asarray(A[:na])[ii], np.asarray(B[:nb])[jj]


def _lib():
    from nextgenmap_amd.pipeline import _lib as load
    lib = load()
    lib.ngm_debug_pair_walk.restype = C.c_int
    lib.ngm_debug_pair_walk.argtypes = [C.c_uint32, C.c_int, C.c_uint32, C.c_int] + [C.c_void_p] * 4 + [C.c_float, C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_uint64] + [C.c_void_p] * 5
    return lib


def run_case(lib, cnt_a, cnt_b, loc, sv, score, rank, cutoff=0.9, min_d=0, max_d=1000, len_a=150, len_b=150):
    n = cnt_a + cnt_b
    loc = np.ascontiguousarray(loc, np.uint32); sv = np.ascontiguousarray(sv, np.uint32); score = np.ascontiguousarray(score, np.float32)
    assert len(loc) == n == len(sv) == len(score)
    rk = None if rank is None else np.ascontiguousarray(rank, np.uint32)
    A, heap_a = reference_order(list(range(cnt_a)), loc, sv, score, rk)
    B, heap_b = reference_order(list(range(cnt_a, n)), loc, sv, score, rk)
    e_ps, e_d, e_a, e_b = expected_walk(A, B, len_a, len_b, loc, score, cutoff, min_d, max_d)
    out_a, out_b = np.zeros(cnt_a, np.uint32), np.zeros(cnt_b, np.uint32)
    mq_a, mq_b, n_combo = C.c_int(-1), C.c_int(-1), C.c_uint64(0)
    cap = len(e_ps) + 16
    c_ps, c_d, c_a, c_b = np.zeros(cap, np.float32), np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    rc = lib.ngm_debug_pair_walk(cnt_a, len_a, cnt_b, len_b, loc.ctypes.data, sv.ctypes.data, score.ctypes.data, None if rk is None else rk.ctypes.data,
                                 cutoff, min_d, max_d, out_a.ctypes.data, out_b.ctypes.data, C.addressof(mq_a), C.addressof(mq_b), cap,
                                 c_ps.ctypes.data, c_d.ctypes.data, c_a.ctypes.data, c_b.ctypes.data, C.addressof(n_combo))
    assert rc == 0, lib.ngm_pipeline_last_error()
    assert out_a.tolist() == A, "list of mate a"
    assert out_b.tolist() == B, "list of mate b"
    assert (mq_a.value, mq_b.value) == (mapq(A, score), mapq(B, score))
    k = int(n_combo.value)
    assert k == len(e_ps), (k, len(e_ps))
    assert np.array_equal(c_a[:k], e_a) and np.array_equal(c_b[:k], e_b), "combinations: which, in which order"
    assert np.array_equal(c_d[:k], e_d) and np.array_equal(c_ps[:k], e_ps)
    return heap_a + heap_b, k


def make_lists(rng, cnt_a, cnt_b, kind, scores="ties"):
    n = cnt_a + cnt_b
    if kind == "satellite":      # every candidate within a few hundred bases of the others: hundreds of mates in every window
        loc = 5_000_000 + rng.integers(0, 3000, n)
    elif kind == "families":     # repeat copies all over the genome, the mates' copies next to each other
        base = rng.integers(10_000, 3_000_000_000, max(cnt_a, cnt_b))
        loc = np.concatenate([base[:cnt_a], base[:cnt_b] + rng.integers(-700, 700, cnt_b)])
    elif kind == "edges":        # locations at both ends of the 32-bit range (the window's bounds are clamped)
        loc = np.where(rng.random(n) < 0.5, rng.integers(0, 1500, n), 0xFFFFFFFF - rng.integers(0, 1500, n))
    else:
        loc = rng.integers(10_000, 3_000_000_000, n)
    sv = rng.integers(0, 1 << 12, n)
    if scores == "ties":
        score = rng.choice([1400.0, 1390.0, 1385.0, 1300.0, 900.0], n, p=[0.35, 0.25, 0.2, 0.1, 0.1])
    elif scores == "equal":
        score = np.full(n, 1234.0)
    elif scores == "distinct":
        score = rng.permutation(n).astype(np.float64) + 5000.0
    elif scores == "few":
        score = rng.integers(1, 4, n).astype(np.float64) * 400.0
    else:
        score = rng.integers(-50, 1500, n).astype(np.float64)
    rank = rng.permutation(1 << 21)[:n] * 2 + (sv & 1)   # 2 x the entering time of the candidate's bin + strand: distinct
    return np.asarray(loc, np.uint64).astype(np.uint32), sv, score, rank


def test_std_sort_restatement_sorts_and_takes_the_heap_path_when_driven_into_it():
    """the restatement on its own: a permutation of its input, ordered -- also through heap sort (a depth limit of 0 is what the
    introsort loop hands over when the limit is spent)"""
    rng = np.random.default_rng(3)
    for n in (1, 2, 16, 17, 100, 1000):
        v = rng.integers(0, 50, n).tolist()
        s = StdSort(lambda a, b: a > b)
        out = s.sort(list(v))
        assert sorted(v, reverse=True) == out
        h = list(v)
        StdSort(lambda a, b: a > b)._heap_sort(h, 0, n)
        assert sorted(v, reverse=True) == h


@pytest.mark.parametrize("kind", ["satellite", "families", "edges", "spread"])
@pytest.mark.parametrize("scores", ["ties", "equal", "distinct", "few", "wide"])
def test_pair_walk_matches_the_reference_algorithm(kind, scores):
    lib = _lib()
    rng = np.random.default_rng(zlib.crc32((kind + "/" + scores).encode()))
    combos = 0
    for cnt_a, cnt_b in ((1, 1), (1, 40), (3, 5), (16, 16), (17, 16), (40, 33), (255, 256), (257, 300), (1200, 900), (2500, 1800)):
        loc, sv, score, rank = make_lists(rng, cnt_a, cnt_b, kind, scores)
        _, k = run_case(lib, cnt_a, cnt_b, loc, sv, score, rank)
        combos += k
    assert combos > 0 or kind == "spread"   # (mates all over a 3 Gbp genome: hardly ever inside one window)


def test_pair_walk_without_ranks_with_repeated_ranks_and_other_windows():
    lib = _lib()
    rng = np.random.default_rng(77)
    loc, sv, score, rank = make_lists(rng, 700, 650, "families", "ties")
    run_case(lib, 700, 650, loc, sv, score, None)                                # no candidate order at all
    rk = rank.copy(); rk[5] = UNKNOWN
    run_case(lib, 700, 650, loc, sv, score, rk)                                  # one candidate's order unknown: as without
    rk = rank.copy(); rk[:700] = rk[:700] // 64 * 64                             # repeated ranks: the comparison path
    run_case(lib, 700, 650, loc, sv, score, rk)
    for cutoff, lo, hi in ((0.9, 0, 1000), (0.5, 100, 400), (1.0, 0, 0), (0.99, 0, 50_000)):   # (max 0: no upper limit)
        loc, sv, score, rank = make_lists(rng, 300, 280, "satellite", "ties")
        run_case(lib, 300, 280, loc, sv, score, rank, cutoff=cutoff, min_d=lo, max_d=hi)
    # sequences that cost a median-of-three quicksort its balance: sorted, reversed, organ pipe, in rank order
    for shape in ("up", "down", "pipe"):
        n = 3000
        base = np.arange(n, dtype=np.float64)
        sc = {"up": base, "down": base[::-1], "pipe": np.minimum(base, n - 1 - base)}[shape] // 3 + 100.0
        order = np.arange(n)
        loc = (1_000_000 + rng.integers(0, 2_000_000, 2 * n)).astype(np.uint32)
        score = np.concatenate([sc, sc])
        rank = np.concatenate([order * 2, order * 2 + 1]).astype(np.uint32)
        heaps, _ = run_case(lib, n, n, loc, np.zeros(2 * n, np.uint32), score, rank, cutoff=0.999)
        assert heaps > 0 or shape != "pipe"   # (the organ pipe spends the depth limit: the heap sort is compared too)


def check_pairs_loop(ps, d, a, b, avg):
    """ScoreBuffer::top1PE's double loop over CheckPairs (src/ScoreBuffer.cpp:397-413, :475-498), on the combinations inside the insert-size
    window in the loop's order; pairDistSum / pairDistCount = avg"""
    top, distance, equal, t1, t2 = np.float32(0.0), 0, 0, -1, -1
    for x in range(len(ps)):
        s, cur = np.float32(ps[x]), int(d[x])
        take = False
        if s > top * np.float32(1.0):
            top, distance, take = s, cur, True
        elif s == top:
            if abs(distance - avg) > abs(cur - avg):
                top, distance, take = s, cur, True
            elif abs(distance) == abs(cur):
                equal += 1
        if take:
            t1, t2 = int(a[x]), int(b[x])
    return [1, t1, t2, equal, distance] if top > 0 else [0, -1, -1, 0, 0]


def test_pair_evaluation_on_all_and_on_the_kept_combinations():
    """eval_pair_seq (the sequential pass's scan) against the reference's loop, on every combination and on the ones the product keeps --
    those that reach the running maximum of the pair score: the others change nothing whatever the running mean"""
    lib = _lib()
    lib.ngm_debug_pair_eval.restype = C.c_int
    lib.ngm_debug_pair_eval.argtypes = [C.c_uint64] + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 2
    rng = np.random.default_rng(20261001)
    kept_total = all_total = 0
    for case in range(400):
        n = int(rng.choice([0, 1, 2, 7, 64, 500, 5000]))
        kind = case % 5
        if kind == 0:
            ps = rng.choice([2800.0, 2790.0, 2785.0, 2700.0], n)                # a repeat family: a handful of pair scores
        elif kind == 1:
            ps = np.full(n, 2468.0)                                             # all equal
        elif kind == 2:
            ps = rng.choice([0.0, -5.0, -20.0], n)                              # nothing positive: no pair
        elif kind == 3:
            ps = np.sort(rng.integers(1, 3000, n)).astype(np.float64)           # a new maximum nearly every time
        else:
            ps = rng.integers(-10, 40, n).astype(np.float64)                    # around zero, many ties
        d = rng.integers(1, 1000, n) if case % 2 else rng.choice([300, 350, 400], n)   # (few insert sizes: the equal-size counter counts)
        ps = np.ascontiguousarray(ps, np.float32); d = np.ascontiguousarray(d, np.int32)
        a = np.ascontiguousarray(rng.integers(0, 1 << 20, n), np.int32); b = np.ascontiguousarray(rng.integers(0, 1 << 20, n), np.int32)
        for avg in (int(rng.integers(150, 600)), 350, 0):
            out_all, out_kept = np.zeros(6, np.int32), np.zeros(6, np.int32)
            rc = lib.ngm_debug_pair_eval(n, ps.ctypes.data, d.ctypes.data, a.ctypes.data, b.ctypes.data, avg, out_all.ctypes.data, out_kept.ctypes.data)
            assert rc == 0, lib.ngm_pipeline_last_error()
            want = check_pairs_loop(ps, d, a, b, avg)
            assert out_all[:5].tolist() == want, (case, n, avg)
            assert out_kept[:5].tolist() == want, (case, n, avg)
            assert out_all[5] == n and out_kept[5] <= n
            kept_total += int(out_kept[5]); all_total += n
    assert 0 < kept_total < all_total   # (the filter does drop combinations in these cases)
