"""-m gpu: the device-resident mapping path (candidate search, gather, score, selection, align).
 * candidate search against the per-read CS log of the REAL reference program (ngm-core-debug built from
   the NextGenMap sources by oracle/ngm_ref.mk, run at test time),
 * gather + score + top-1/MAPQ + align + final position against the C oracle composed with a Python
   restatement of ScoreBuffer/AlignmentBuffer's host glue (tests/ngm_host_model.py)."""
import os
import re

import numpy as np
import pytest

import ngm_host_model as HM
import oracle_lib as O
import ref_files as RF
import simulate as S

pytestmark = pytest.mark.gpu


def _world(tmp_path, n_reads=3000, read_len=100, seed=5):
    contigs = S.make_genome([250000, 180001, 90000], seed=seed, repeat_families=10, repeat_len=500, copies=8, divergence=0.01)
    fa = str(tmp_path / "ref.fa")
    with open(fa, "wb") as f:
        for i, g in enumerate(contigs):
            f.write(b">chr%d\n" % (i + 1))
            b = g.tobytes()
            for o in range(0, len(b), 70):
                f.write(b[o:o + 70] + b"\n")
    reads = S.make_reads(contigs, n_reads, read_len, seed=seed + 1, sub_rate=0.02, indel_rate=0.003)
    # a few awkward reads: N-rich, N runs near the end, all N, low complexity, very short
    rng = np.random.default_rng(seed)
    for j in range(40):
        name, seq, q = reads[j]
        seq = seq.copy()
        if j % 5 == 0:
            seq[rng.integers(0, len(seq), 8)] = ord("N")
        elif j % 5 == 1:
            seq[-16:-13] = ord("N")
        elif j % 5 == 2:
            seq[-15:-13] = ord("N")
        elif j % 5 == 3:
            seq[:] = np.resize(np.frombuffer(b"AC", np.uint8), len(seq))
        else:
            seq = seq[:20 + j]
        reads[j] = (name, seq, q[:len(seq)])
    reads[40] = (reads[40][0], np.full(read_len, ord("N"), np.uint8), reads[40][2])
    fq = str(tmp_path / "reads.fq")
    S.write_fastq(fq, reads)
    return contigs, reads, fa, fq


@pytest.mark.skipif(not RF.have_reference_binary(), reason="reference binary not built (oracle/ngm_ref.mk)")
@pytest.mark.parametrize("fast_items", [12, 24], ids=["12-items-per-lane", "24-items-per-lane"])
def test_candidate_search_matches_reference_program(tmp_path, monkeypatch, fast_items):
    monkeypatch.setenv("NGM_HIP_CS_FAST_ITEMS", str(fast_items))  # both instantiations of the fast kernel
    from nextgenmap_amd.pipeline import Mapper, Reference
    contigs, reads, fa, fq = _world(tmp_path)
    r = RF.run_ngm(["-r", fa, "-q", fq, "-o", str(tmp_path / "out.sam"), "--affine", "-t", "1", "--no-progress", "-s", "0.5",
                    "--log", "--log-lvl", "8200"], debug=True, cwd=str(tmp_path))
    assert "Done" in (r.stdout + r.stderr), (r.stdout + r.stderr)[-3000:]
    want = {}
    summ = {}
    for line in r.stdout.splitlines():
        m = re.match(r"8192\tREAD_(\d+)\tCS_RESULTS\tInternal location: (\d+) \(([+-])\).*Score: ([0-9.]+) \(ACCEPT\)", line)
        if m:
            want.setdefault(int(m.group(1)), set()).add((int(m.group(2)), 0 if m.group(3) == "+" else 1, float(m.group(4))))
            continue
        m = re.match(r"8\tREAD_(\d+)\tCS\tSummary: (\d+) CMRs \((\d+) without filter\), (\d+) accepted, theta: ([0-9.]+), max: ([0-9.]+)", line)
        if m:  # "accepted" = strand scores at or above the final threshold = the candidates CollectResultsStd emits
            summ[int(m.group(1))] = (int(m.group(4)), float(m.group(5)), float(m.group(6)))
    assert len(summ) == len(reads)
    mq = re.search(r"Max\. k-mer frequency set so (\d+)", r.stdout + r.stderr)
    qlen = (max(len(s) for _, s, _ in reads) | 1) + 1

    ref = Reference.from_fasta(fa)
    assert int(mq.group(1)) == ref.auto_max_kfreq
    mp = Mapper(ref, qlen, 20, sensitivity=0.5)
    rows = Mapper.reads_to_rows([s for _, s, _ in reads], qlen)
    offs, maxv, loc, strand, votes = mp.candidate_search(rows)
    bad = 0
    for i in range(len(reads)):
        got = {(int(loc[j]) >> 2, int(strand[j]), float(votes[j])) for j in range(offs[i], offs[i + 1])}
        exp = want.get(i, set())
        assert all((int(loc[j]) & 3) == 2 for j in range(offs[i], offs[i + 1]))
        if got != exp or maxv[i] != summ[i][2] or len(got) != summ[i][0]:
            bad += 1
            if bad < 5:
                print("read", i, "got", sorted(got)[:6], "want", sorted(exp)[:6], maxv[i], summ[i])
    assert bad == 0, "%d reads with a different candidate set / max votes" % bad
    mp.close(); ref.close()


def test_window_decode_on_device_matches_host_model(tmp_path):
    from nextgenmap_amd.pipeline import Reference
    contigs = S.make_genome([5000, 3001], seed=9, repeat_families=1, copies=1, n_runs=1)
    ref = Reference.from_contigs(contigs)
    g, geom, n_bases = HM.concat_genome(contigs)
    assert n_bases - 1 == ref.concat_len
    rng = np.random.default_rng(1)
    offsets = list(rng.integers(900, n_bases - 200, 60)) + [n_bases - 130, n_bases - 125, n_bases - 5, n_bases - 2, n_bases - 1, n_bases + 7]
    for off in offsets:
        for blen in (124, 180, 179, 122, 295):
            ok_m, exp = HM.decode_ref(g, n_bases, int(off), blen)
            ok, got = ref.decode(int(off), blen)
            assert ok == ok_m
            if ok:
                assert got == bytes(exp), (off, blen, got[-12:], bytes(exp)[-12:])
    ref.close()


@pytest.mark.parametrize("mode", [0, 1], ids=["local", "endfree"])
def test_map_se_matches_oracle_composition(tmp_path, mode):
    from nextgenmap_amd.pipeline import Mapper, Reference
    contigs, reads, fa, fq = _world(tmp_path, n_reads=1500, seed=11)
    q, c = 102, 20
    ref = Reference.from_contigs(contigs)
    mp = Mapper(ref, q, c, sensitivity=0.5, mode=mode)
    rows = Mapper.reads_to_rows([s for _, s, _ in reads], q)
    offs, maxv, loc, strand, votes = mp.candidate_search(rows)
    hits, cig, md = mp.map_se(rows)
    g, geom, n_bases = HM.concat_genome(contigs)
    n_mapped = 0
    for i in range(len(reads)):
        nc = offs[i + 1] - offs[i]
        h = hits[i]
        assert h["n_candidates"] == nc and h["max_votes"] == maxv[i]
        if nc == 0:
            assert not h["mapped"]
            continue
        L = int(np.count_nonzero(rows[i]))
        wins, qrys, keys = [], [], []
        for j in range(offs[i], offs[i + 1]):
            ok, w = HM.decode_ref(g, n_bases, int(loc[j]) - (c >> 1), ((q + c) | 1) + 1)
            assert ok
            wins.append(w[:q + c])
            qrys.append(HM.revcomp_row(rows[i], L) if strand[j] else rows[i])
            keys.append((int(loc[j]) << 1) | int(strand[j]))
        sc = O.oracle_score(mode, np.array(wins), np.array(qrys), c)
        w, mq, num, best = HM.top1(list(sc), keys)
        assert (h["mapq"], h["n_best"]) == (mq, num), (i, h, mq, num, sc)
        assert h["score"] == (best if best > 0 else sc[w])
        # several candidates with the best score: the product keeps the first one in the reference's candidate order
        # (tests/test_gpu_cli.py checks that order against the real program); here any of the tied ones is consistent
        tied = [w] if num == 1 else [t for t in range(nc) if (sc[t] == best or not best > 0)]

        def expect(w):
            j = offs[i] + w
            ok, aw = HM.decode_ref(g, n_bases, int(loc[j]) - (c >> 1), (q + c) | 2)
            res, ocig, omd = O.oracle_align(mode, aw[None, :q + c], qrys[w][None, :], c)
            if not res["ok"][0]:
                return None
            conv = HM.convert(geom, int(loc[j]) + int(res["position_offset"][0]) - (c >> 1))
            if conv is None:
                return None
            return (conv[0], conv[1], int(strand[j]), ocig[0], omd[0], int(res["nm"][0]), int(res["qstart"][0]), int(res["qend"][0]),
                    np.float32(res["identity"][0]))
        wants = [expect(t) for t in tied]
        if not h["mapped"]:
            assert None in wants, (i, h, wants)
            continue
        got = (h["contig"], h["pos"], h["reverse"], cig[i], md[i], h["nm"], h["qstart"], h["qend"], np.float32(h["identity"]))
        assert got in [x for x in wants if x is not None], (i, got, wants)
        n_mapped += 1
    assert n_mapped > 0.9 * len(reads)
    # truth check: simulated origin within the band of the reported position
    near = 0
    for i, (name, s, _) in enumerate(reads[41:], start=41):
        _, ci, p, st = name.split("_")
        if hits[i]["mapped"] and hits[i]["contig"] == int(ci) and abs(int(hits[i]["pos"]) - int(p)) <= 12:
            near += 1
    assert near > 0.93 * (len(reads) - 41)
    mp.close(); ref.close()


def test_host_threads_are_pinned_to_the_gpus_numa_node():
    """ngm_host_pin_to_device_node: 0 (no NUMA information / not permitted) or the size of the CPU set the calling thread now has."""
    import ctypes as C
    import os
    import threading
    from nextgenmap_amd.engine import load_library
    lib = load_library()
    lib.ngm_host_pin_to_device_node.restype = C.c_int
    out = {}

    def probe():  # (in a thread of its own: the affinity of the test runner stays what it was)
        out["n"] = lib.ngm_host_pin_to_device_node(0)
        out["cpus"] = len(os.sched_getaffinity(0))
    t = threading.Thread(target=probe)
    t.start()
    t.join()
    assert out["n"] >= 0
    if out["n"] > 0:
        assert out["cpus"] == out["n"]
