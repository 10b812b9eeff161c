"""-m gpu: `ngm-hip --bam` against the BAM file the REAL reference program writes (`ngm --affine --bam`, bamtools 2.3.0 behind
src/writer/BAMWriter.cpp:147-460): both files are BGZF-decoded and parsed here, and every record must be equal in every field --
core (refID, pos, bin, MAPQ, flag, mate fields, TLEN), read name, packed CIGAR, 4-bit sequence, qualities, and the tag block byte
for byte (AS NM NH XI X0 XE XR MD [RG]); the header text up to the program line; the reference dictionary."""
import gzip
import os
import struct
import subprocess

import numpy as np
import pytest

import ref_files as RF
import simulate as S

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "nextgenmap_amd", "ngm-hip")


def decode_bam(path):
    raw = open(path, "rb").read()
    assert raw[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"), "BGZF end-of-file block missing"
    data = gzip.decompress(raw)
    assert data[:4] == b"BAM\1"
    l_text, = struct.unpack_from("<i", data, 4)
    text = data[8:8 + l_text].decode()
    at = 8 + l_text
    n_ref, = struct.unpack_from("<i", data, at)
    at += 4
    refs = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", data, at)
        name = data[at + 4:at + 4 + l_name - 1].decode()
        l_ref, = struct.unpack_from("<i", data, at + 4 + l_name)
        refs.append((name, l_ref))
        at += 8 + l_name
    recs = []
    while at < len(data):
        block, = struct.unpack_from("<i", data, at)
        ref_id, pos, bin_mq_nl, flag_nc, l_seq, mate_ref, mate_pos, tlen = struct.unpack_from("<iiIIiiii", data, at + 4)
        l_name, n_cig = bin_mq_nl & 0xFF, flag_nc & 0xFFFF
        p = at + 36
        name = data[p:p + l_name - 1]; p += l_name
        cigar = data[p:p + 4 * n_cig]; p += 4 * n_cig
        seq = data[p:p + (l_seq + 1) // 2]; p += (l_seq + 1) // 2
        qual = data[p:p + l_seq]; p += l_seq
        tags = data[p:at + 4 + block]
        recs.append(dict(name=name, ref_id=ref_id, pos=pos, bin=bin_mq_nl >> 16, mapq=(bin_mq_nl >> 8) & 0xFF, flag=flag_nc >> 16, cigar=cigar, l_seq=l_seq,
                         seq=seq, qual=qual, mate_ref=mate_ref, mate_pos=mate_pos, tlen=tlen, tags=tags))
        at += 4 + block
    return text, refs, recs


def _case(tmp_path, paired):
    contigs = S.make_genome([400000, 300001], seed=801, repeat_families=10, repeat_len=500, copies=6)
    fa = str(tmp_path / "ref.fa")
    with open(fa, "wb") as f:
        for i, g in enumerate(contigs):
            f.write(b">chr%d\n" % (i + 1))
            b = g.tobytes()
            for o in range(0, len(b), 70):
                f.write(b[o:o + 70] + b"\n")
    fq = str(tmp_path / "reads.fq")
    rng = np.random.default_rng(8)
    if paired:
        r1, r2 = S.make_reads(contigs, 2500, 100, seed=802, sub_rate=0.02, indel_rate=0.003, paired=True)
        for k in range(0, 60, 3):   # pairs that cannot be proper / mates that do not map
            r2[k] = (r2[k][0], S.ACGT[rng.integers(0, 4, 100)], r2[k][2])
        for k in range(1, 60, 3):
            r2[k] = (r2[k][0], r2[k + 300][1], r2[k][2])
        S.write_fastq(fq, [x for pair in zip(r1, r2) for x in pair])
        return fa, ["-p", "-q", fq]
    reads = S.make_reads(contigs, 4000, 100, seed=803, sub_rate=0.02, indel_rate=0.003)
    for k in range(0, 40, 2):
        reads[k] = (reads[k][0], S.ACGT[rng.integers(0, 4, 100)], reads[k][2])
    S.write_fastq(fq, reads)
    return fa, ["-q", fq]


@pytest.mark.skipif(not RF.have_reference_binary(), reason="reference binary not built (oracle/ngm_ref.mk)")
@pytest.mark.parametrize("layout,extra", [("se", []), ("pe", []), ("se", ["--hard-clip", "--rg-id", "g1", "--rg-sm", "s1"]), ("se", ["-n", "3"])],
                         ids=["single-end", "paired-end", "hard-clip-read-group", "top3"])
def test_bam_equals_reference_program(tmp_path, layout, extra):
    from nextgenmap_amd import build
    build.build()
    fa, inp = _case(tmp_path, layout == "pe")
    d1 = tmp_path / "refrun"
    d1.mkdir()
    fa1 = str(d1 / "ref.fa")
    os.link(fa, fa1)
    r = RF.run_ngm(["-r", fa1, "-o", str(d1 / "out.bam"), "--affine", "--bam", "-t", "1", "--no-progress"] + inp + extra, cwd=str(d1))
    assert "Done" in (r.stdout + r.stderr), (r.stdout + r.stderr)[-1500:]
    c = subprocess.run([CLI, "-r", fa, "-o", str(tmp_path / "hip.bam"), "--affine", "--bam"] + inp + extra, capture_output=True, text=True)
    assert c.returncode == 0, c.stderr[-2000:]
    ta, ra, a = decode_bam(str(d1 / "out.bam"))
    tb, rb, b = decode_bam(str(tmp_path / "hip.bam"))
    assert ra == rb
    strip = lambda t: [l.split("\tCL:")[0] if l.startswith("@PG") else l for l in t.splitlines()]
    assert strip(ta) == strip(tb), (ta, tb)
    assert len(a) == len(b) and len(a) >= 4000
    # the reference writes reads without candidates as soon as the search is done and the others after the alignment stage
    # (src/CS.cpp, src/AlignmentBuffer.cpp): its record ORDER is not the input order; compare per read (and mate)
    def by_read(recs):
        out = {}
        for x in recs:
            out.setdefault((x["name"], x["flag"] & 0xC0), []).append(tuple(sorted((k, v) for k, v in x.items())))
        return {k: sorted(v) for k, v in out.items()}
    da, db = by_read(a), by_read(b)
    assert set(da) == set(db)
    diff = [(da[k], db[k]) for k in da if da[k] != db[k]]
    print("reads differing:", len(diff), "of", len(da), "; unmapped:", sum(1 for x in a if x["flag"] & 4))
    for x, y in diff[:3]:
        print([(u, v) for u, v in zip(x[0], y[0]) if u != v])
    assert not diff


def test_bam_shards_concatenate(tmp_path):
    """`--bam --shard i/N`: whole BGZF blocks per shard, the header with shard 0, the end-of-file block with the last one -- the
    concatenation is one valid BAM file with the records of the unsharded run."""
    fa, inp = _case(tmp_path, False)
    one = str(tmp_path / "one.bam")
    c = subprocess.run([CLI, "-r", fa, "-o", one, "--affine", "--bam", "--batch-size", "1024"] + inp, capture_output=True, text=True)
    assert c.returncode == 0, c.stderr[-2000:]
    parts = b""
    for i in range(2):
        out = str(tmp_path / ("part%d.bam" % i))
        c = subprocess.run([CLI, "-r", fa, "-o", out, "--affine", "--bam", "--batch-size", "1024", "--shard", "%d/2" % i] + inp, capture_output=True, text=True)
        assert c.returncode == 0, c.stderr[-2000:]
        parts += open(out, "rb").read()
    cat = str(tmp_path / "cat.bam")
    open(cat, "wb").write(parts)
    ta, ra, a = decode_bam(one)
    tb, rb, b = decode_bam(cat)
    assert ra == rb and len(a) == len(b) == 4000
    strip = lambda t: [l.split("\tCL:")[0] if l.startswith("@PG") else l for l in t.splitlines()]
    assert strip(ta) == strip(tb)
    assert a == b


@pytest.mark.parametrize("layout,extra", [("se", []), ("pe", []), ("pe", ["--silent-clip", "--no-unal", "--rg-id", "x", "-i", "0.9"]), ("se", ["--bs-mapping"]),
                                          ("pe", ["--hard-clip", "-R", "0.7", "-Q", "10"])],
                         ids=["single-end", "paired-end", "pe-silent-clip-no-unal-rg-identity", "se-bisulfite", "pe-hard-clip-filters"])
def test_bam_paths_write_the_same_records(tmp_path, layout, extra):
    """the three ways `ngm-hip --bam` can make its file -- records and BGZF blocks on the GPU (default), records on the host and blocks on
    the GPU, records on the host and zlib level 6 -- decode to the same header and the same records in the same order"""
    fa, inp = _case(tmp_path, layout == "pe")
    decoded = []
    for tag, env in (("gpu", {}), ("host_records", {"NGM_HIP_BAM_HOST_RECORDS": "1"}), ("zlib", {"NGM_HIP_BAM_ZLIB": "1"})):
        out = str(tmp_path / (tag + ".bam"))
        pers = [] if "--bs-mapping" in extra else ["--affine"]   # (bisulfite mapping is the default personality's)
        c = subprocess.run([CLI, "-r", fa, "-o", out, "--bam", "--batch-size", "1500"] + pers + inp + extra, capture_output=True, text=True, env=dict(os.environ, **env))
        assert c.returncode == 0, c.stderr[-2000:]
        assert ("BAM records and their BGZF blocks written on the GPU" in c.stderr) == (tag == "gpu"), c.stderr[-1500:]
        decoded.append(decode_bam(out))
    (t0, r0, a), (t1, r1, b), (t2, r2, z) = decoded
    strip = lambda t: [l.split("\tCL:")[0] if l.startswith("@PG") else l for l in t.splitlines()]   # (the command line names the output file)
    assert strip(t0) == strip(t1) == strip(t2) and r0 == r1 == r2
    assert len(a) == len(b) == len(z) and len(a) > 1000
    bad = [i for i in range(len(a)) if a[i] != z[i]]
    for i in bad[:3]:
        print([(k, a[i][k], z[i][k]) for k in a[i] if a[i][k] != z[i][k]])
    assert not bad and b == z
