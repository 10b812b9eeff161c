"""-m gpu: the HBM-resident reference + k-mer index against the index the REAL reference program
(oracle/_ref/ngm/ngm-core, built from the NextGenMap sources) writes for the same FASTA."""
import os

import numpy as np
import pytest

import ref_files as RF
import simulate as S

pytestmark = pytest.mark.gpu


def _tricky_genome():
    contigs = S.make_genome([60000, 35001, 20000], seed=77, repeat_families=4, repeat_len=300, copies=5, n_runs=4)
    rng = np.random.default_rng(3)
    g = contigs[0]
    g[5000:5400] = ord("A")                      # homopolymer: consecutive identical indexed k-mers in one bin
    g[7000:7300] = np.resize(np.frombuffer(b"ACG", np.uint8), 300)  # period-3 repeat
    g[9000] = ord("N")                           # single N
    g[9100:9102] = ord("N")                      # two Ns
    g[0:3] = ord("N")                            # contig starting with Ns
    contigs[1][-20:-13] = ord("N")               # N run followed by exactly 13 bases at the contig end
    contigs[2][-16:-14] = ord("N")               # ... followed by 14
    contigs[2][100:140] = np.frombuffer(b"acgtRYKMnnacgtacgtacgtacgtacgtacgtacgtac", np.uint8)  # lower case / IUPAC
    contigs.append(S.ACGT[rng.integers(0, 4, 9)])   # too short: skipped by the reference
    contigs.append(S.ACGT[rng.integers(0, 4, 501)])
    return contigs


@pytest.mark.skipif(not RF.have_reference_binary(), reason="reference binary not built (oracle/ngm_ref.mk)")
def test_index_matches_the_reference_programs_index(tmp_path, monkeypatch):
    from nextgenmap_amd.pipeline import Reference
    monkeypatch.setenv("NGM_HIP_NO_CACHE", "1")  # build from the FASTA, do not load what the reference program wrote
    contigs = _tricky_genome()
    fa = str(tmp_path / "ref.fa")
    with open(fa, "wb") as f:
        for i, g in enumerate(contigs):
            f.write(b">chr%d some description\n" % (i + 1))
            b = g.tobytes()
            for o in range(0, len(b), 61):
                f.write(b[o:o + 61] + b"\n")
    r = RF.run_ngm(["-r", fa], cwd=str(tmp_path))
    assert os.path.exists(fa + "-ht-13-2.3.ngm"), r.stdout + r.stderr
    ht = RF.read_ht_file(fa + "-ht-13-2.3.ngm")
    enc = RF.read_enc_file(fa + "-enc.2.ngm")

    ref = Reference.from_fasta(fa)
    # genome geometry
    ours = ref.contigs
    assert [c[0].encode() for c in ours] == [bytes(n) for n in enc["refidx"]["name"]]
    assert [c[1] for c in ours] == [int(x) for x in enc["refidx"]["SeqStart"]]
    assert [c[2] for c in ours] == [int(x) for x in enc["refidx"]["SeqLen"]]
    assert ref.concat_len == enc["n_bases"] - 1
    # index: per-k-mer list lengths (raw and as a lookup sees them) and every stored position
    counts, raw, pos = ref.index_copy()
    assert np.array_equal(raw, ht["raw_counts"])
    assert np.array_equal(counts, ht["counts"])
    used = ht["counts"] > 0
    # positions of used k-mers, k-mer by k-mer (unused k-mers keep zeroed slots in the reference's table)
    our_starts = np.concatenate([[0], np.cumsum(raw.astype(np.int64))])[:-1]
    kk = np.nonzero(used)[0]
    ref_pos = np.concatenate([ht["positions"][ht["starts"][k]:ht["starts"][k] + ht["counts"][k]] for k in kk])
    our_pos = np.concatenate([pos[our_starts[k]:our_starts[k] + counts[k]] for k in kk])
    assert np.array_equal(ref_pos, our_pos)
    # auto max k-mer frequency: the reference prints it
    import re
    m = re.search(r"Max\. k-mer frequency set so (\d+)", r.stdout + r.stderr)
    assert m and int(m.group(1)) == ref.auto_max_kfreq
    # from_contigs gives the same thing as from_fasta
    ref2 = Reference.from_contigs(contigs)
    c2, r2, p2 = ref2.index_copy()
    assert np.array_equal(c2, counts) and np.array_equal(p2, pos)
    # convert(): spacer rejection and contig mapping
    for name, start, ln in ours:
        assert ref.convert(start) == (ours.index((name, start, ln)), 0)
        assert ref.convert(start + ln - 1) == (ours.index((name, start, ln)), ln - 1)
    assert ref.convert(ours[1][1] - 5) is None
    ref.close(); ref2.close()


@pytest.mark.skipif(not RF.have_reference_binary(), reason="reference binary not built (oracle/ngm_ref.mk)")
def test_reference_program_accepts_our_cache_files(tmp_path):
    """Write <ref>-enc.2.ngm / <ref>-ht-13-2.3.ngm from the GPU-built index: byte-compare with the files the
    reference writes itself, and check that the reference program maps identically when it loads ours."""
    from nextgenmap_amd.pipeline import Reference
    contigs = _tricky_genome()
    d1, d2 = tmp_path / "theirs", tmp_path / "ours"
    d1.mkdir(); d2.mkdir()
    for d in (d1, d2):
        with open(str(d / "ref.fa"), "wb") as f:
            for i, g in enumerate(contigs):
                f.write(b">chr%d\n" % (i + 1))
                b = g.tobytes()
                for o in range(0, len(b), 60):
                    f.write(b[o:o + 60] + b"\n")
    reads = S.make_reads([c for c in contigs if len(c) > 1000], 400, 100, seed=4)
    fq = str(tmp_path / "r.fq")
    S.write_fastq(fq, reads)
    ref = Reference.from_fasta(str(d2 / "ref.fa"))
    ref.write_ngm_cache(str(d2 / "ref.fa"))
    ref.close()
    outs = []
    for d in (d1, d2):
        r = RF.run_ngm(["-r", str(d / "ref.fa"), "-q", fq, "-o", str(d / "out.sam"), "--affine", "-t", "1", "--no-progress", "-s", "0.5"], cwd=str(d))
        assert "Done" in r.stdout + r.stderr, (r.stdout + r.stderr)[-2000:]
        if d is d2:
            assert "Reading RefTable from" in r.stdout + r.stderr and "Building reference table" not in r.stdout + r.stderr
        outs.append(sorted(l for l in open(str(d / "out.sam")) if not l.startswith("@")))
    assert outs[0] == outs[1] and len(outs[0]) == 400
    ht1, ht2 = RF.read_ht_file(str(d1 / "ref.fa-ht-13-2.3.ngm")), RF.read_ht_file(str(d2 / "ref.fa-ht-13-2.3.ngm"))
    assert np.array_equal(ht1["counts"], ht2["counts"]) and np.array_equal(ht1["positions"], ht2["positions"])
    e1, e2 = RF.read_enc_file(str(d1 / "ref.fa-enc.2.ngm")), RF.read_enc_file(str(d2 / "ref.fa-enc.2.ngm"))
    nb = e1["n_bases"] // 2
    assert e1["n_bases"] == e2["n_bases"] and np.array_equal(e1["data"][:nb], e2["data"][:nb])


@pytest.mark.skipif(not RF.have_reference_binary(), reason="reference binary not built (oracle/ngm_ref.mk)")
def test_loads_the_reference_programs_cache_files(tmp_path, monkeypatch):
    """An index written by NextGenMap itself drops in: <fasta>-enc.2.ngm + <fasta>-ht-13-2.3.ngm are loaded instead of
    rebuilding, and give the same reference, index, windows and mappings as our own build."""
    from nextgenmap_amd.pipeline import Mapper, Reference
    contigs = _tricky_genome()
    fa = str(tmp_path / "ref.fa")
    with open(fa, "wb") as f:
        for i, g in enumerate(contigs):
            f.write(b">chr%d some description\n" % (i + 1))
            b = g.tobytes()
            for o in range(0, len(b), 61):
                f.write(b[o:o + 61] + b"\n")
    r = RF.run_ngm(["-r", fa], cwd=str(tmp_path))
    assert os.path.exists(fa + "-ht-13-2.3.ngm"), r.stdout + r.stderr
    cached = Reference.from_cache(fa)
    also = Reference.from_fasta(fa)            # finds the cache, like the reference program does
    monkeypatch.setenv("NGM_HIP_NO_CACHE", "1")
    built = Reference.from_fasta(fa)
    for other in (cached, also):
        assert other.contigs == built.contigs and other.concat_len == built.concat_len
        assert other.auto_max_kfreq == built.auto_max_kfreq and other.index_entries == built.index_entries
        c1, r1, p1 = other.index_copy()
        c2, r2, p2 = built.index_copy()
        assert np.array_equal(c1, c2) and np.array_equal(r1, r2)
        starts = np.concatenate([[0], np.cumsum(r1.astype(np.int64))])[:-1]
        for k in np.nonzero(c1 > 0)[0][::97]:
            assert np.array_equal(p1[starts[k]:starts[k] + c1[k]], p2[starts[k]:starts[k] + c2[k]])
    for off in (0, 1, 999, 1000, 1001, 5003, 60990, 61999, 62003, built.concat_len - 50, built.concat_len - 1):
        assert cached.decode(off, 180) == built.decode(off, 180), off
    reads = S.make_reads(contigs[:3], 400, 100, seed=5)
    rows = Mapper.reads_to_rows([x[1] for x in reads], 102)
    m1, m2 = Mapper(cached, 102, 20), Mapper(built, 102, 20)
    h1, c1, d1 = m1.map_se(rows)
    h2, c2, d2 = m2.map_se(rows)
    assert np.array_equal(h1, h2) and c1 == c2 and d1 == d2
    m1.close(); m2.close(); cached.close(); also.close(); built.close()
    # wrong parameters are refused, not silently used
    with pytest.raises(Exception):
        Reference.from_cache(fa, kmer=12)
