"""-m gpu: the REAL NextGenMap program with libngm_hip.so plugged in behind IAlignment (oracle/build_dropin.sh: the
reference's own translation units, _NGM::CreateAlignment patched as INTEGRATION.md section A says) against the stock program.
This is the drop-in claim itself: the reference's CS / ScoreBuffer / AlignmentBuffer / SAMWriter drive BatchScore / BatchAlign
of this library through the reference's own vtable calls, and the SAM file must not change."""
import os
import subprocess

import numpy as np
import pytest

import ref_files as RF
import simulate as S

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "oracle", "_ref", "dropin", "ngm-core-hip")
CLI = os.path.join(ROOT, "nextgenmap_amd", "ngm-hip")
needs = pytest.mark.skipif(not (RF.have_reference_binary() and os.path.exists(DROPIN)), reason="oracle/_ref/ngm/ngm-core or oracle/_ref/dropin/ngm-core-hip not built")


def _case(tmp_path, paired):
    contigs = S.make_genome([400000, 300001], seed=901, repeat_families=10, repeat_len=500, copies=6)
    fa = str(tmp_path / "ref.fa")
    with open(fa, "wb") as f:
        for i, g in enumerate(contigs):
            f.write(b">chr%d\n" % (i + 1))
            b = g.tobytes()
            for o in range(0, len(b), 70):
                f.write(b[o:o + 70] + b"\n")
    fq = str(tmp_path / "reads.fq")
    if paired:
        r1, r2 = S.make_reads(contigs, 2500, 100, seed=902, sub_rate=0.02, indel_rate=0.003, paired=True)
        S.write_fastq(fq, [x for pair in zip(r1, r2) for x in pair])
        return fa, ["-p", "-q", fq], 5000
    S.write_fastq(fq, S.make_reads(contigs, 4000, 100, seed=903, sub_rate=0.02, indel_rate=0.003))
    return fa, ["-q", fq], 4000


def _body(path):
    return [l for l in open(path) if not l.startswith("@PG")]


def _run(binary, fa, args, out, cwd):
    r = subprocess.run([binary, "-r", fa, "-o", out, "-t", "1", "--no-progress"] + args, capture_output=True, text=True, cwd=cwd, timeout=1800)
    assert "Done" in (r.stdout + r.stderr), (r.stdout + r.stderr)[-2500:]
    return r.stdout + r.stderr


@needs
@pytest.mark.parametrize("layout", ["single-end", "paired-end", "single-end-end-to-end"])
def test_real_ngm_with_hip_plugin_writes_the_same_sam(tmp_path, layout):
    fa, inp, n = _case(tmp_path, layout == "paired-end")
    extra = ["-e"] if layout.endswith("end-to-end") else []
    stock, plug = str(tmp_path / "stock.sam"), str(tmp_path / "plugin.sam")
    _run(RF.NGM_CORE, fa, inp + ["--affine"] + extra, stock, str(tmp_path))
    log = _run(DROPIN, fa, inp + ["--affine"] + extra, plug, str(tmp_path))
    a, b = _body(stock), _body(plug)
    assert len([l for l in a if not l.startswith("@")]) == n
    diff = [(x, y) for x, y in zip(a, b) if x != y]
    print("lines differing:", len(diff), "of", len(a))
    assert len(a) == len(b) and not diff, str(diff[:2])[:1200]


@needs
def test_real_ngm_with_hip_plugin_default_personality_equals_ngm_hip(tmp_path):
    """NGM's default (linear-gap, OpenCL) personality cannot run in the stock build here (no OpenCL device); through the plugin
    it runs on the MI355X.  Its SAM records must equal what the ngm-hip command line writes for the same input: two
    independent hosts (the reference's own pipeline vs. this repository's device-resident pipeline) above the same kernels."""
    fa, inp, n = _case(tmp_path, False)
    plug, ours = str(tmp_path / "plugin.sam"), str(tmp_path / "ours.sam")
    log_plug = _run(DROPIN, fa, inp, plug, str(tmp_path))
    c = subprocess.run([CLI, "-r", fa, "-o", ours] + inp, capture_output=True, text=True)
    print("\n".join(l for l in log_plug.splitlines() if "ensitivity" in l or "orridor" in l or "read length" in l))
    print("\n".join(l for l in c.stderr.splitlines() if "ensitivity" in l or "orridor" in l or "read length" in l))
    assert c.returncode == 0, c.stderr[-2000:]
    rec = lambda p: {l.split("\t", 1)[0]: l for l in open(p) if not l.startswith("@")}
    a, b = rec(plug), rec(ours)
    assert set(a) == set(b) and len(a) == n
    diff = [(a[k], b[k]) for k in a if a[k] != b[k]]
    print("records differing:", len(diff), "of", len(a))
    assert not diff, str(diff[:2])[:1200]


@needs
def test_default_personality_paired_end_real_program_vs_ngm_hip(tmp_path):
    """Linear (default) personality, paired-end: the real program driving this library through IAlignment (its own top1PE /
    CheckPairs / SAMWriter::DoWritePair above BatchScore / BatchAlign) against the device-resident pipeline of ngm-hip."""
    fa, inp, n = _case(tmp_path, True)
    plug, ours = str(tmp_path / "plugin.sam"), str(tmp_path / "ours.sam")
    _run(DROPIN, fa, inp, plug, str(tmp_path))
    c = subprocess.run([CLI, "-r", fa, "-o", ours] + inp, capture_output=True, text=True)
    assert c.returncode == 0, c.stderr[-2000:]
    rec = lambda p: {(l.split("\t", 2)[0], int(l.split("\t", 2)[1]) & 0xC0): l for l in open(p) if not l.startswith("@")}
    a, b = rec(plug), rec(ours)
    assert set(a) == set(b) and len(a) == n
    diff = [(a[k], b[k]) for k in a if a[k] != b[k]]
    print("records differing:", len(diff), "of", len(a))
    assert not diff, str(diff[:2])[:1500]


def test_two_device_entries_on_one_gpu_give_the_same_output(tmp_path):
    """`-g 0,0`: two references and two sets of mappers (the multi-GPU code path: one reference per listed device, workers
    spread over them, one shared paired-end state, one ordered writer) -- on the one GPU of the test box.  Output must equal the
    single-device run byte for byte."""
    fa, inp, n = _case(tmp_path, True)
    outs = []
    for tag, extra in (("one", []), ("two", ["-g", "0,0", "--batch-size", "1024"])):
        out = str(tmp_path / (tag + ".sam"))
        c = subprocess.run([CLI, "-r", fa, "-o", out, "--affine"] + inp + extra, capture_output=True, text=True)
        assert c.returncode == 0, c.stderr[-2000:]
        if extra:
            assert "2 GPU(s)" in c.stderr
        outs.append([l for l in open(out) if not l.startswith("@PG")])
    assert outs[0] == outs[1] and len([l for l in outs[0] if not l.startswith("@")]) == n


@pytest.mark.parametrize("layout", ["single-end", "paired-end"])
def test_shards_concatenate_to_the_single_run(tmp_path, layout):
    """`--shard i/N` maps the i-th contiguous range of the input and writes its records (shard 0: with the header);
    `-g a,b --shard-output` runs one such process per listed GPU and appends the pieces in shard order.  Single-end output must
    equal the unsharded run byte for byte; paired-end shards restart the running mean insert size (ScoreBuffer.cpp:420-422 -- the
    documented tolerance: only equal-score pair ties may differ), so there the records are compared one by one."""
    paired = layout == "paired-end"
    fa, inp, n = _case(tmp_path, paired)
    one = str(tmp_path / "one.sam")
    c = subprocess.run([CLI, "-r", fa, "-o", one, "--batch-size", "1024"] + inp, capture_output=True, text=True)
    assert c.returncode == 0, c.stderr[-2000:]
    parts = []
    for i in range(3):
        out = str(tmp_path / ("part%d.sam" % i))
        c = subprocess.run([CLI, "-r", fa, "-o", out, "--batch-size", "1024", "--shard", "%d/3" % i] + inp, capture_output=True, text=True)
        assert c.returncode == 0, c.stderr[-2000:]
        parts.append(open(out).read())
    assert parts[0].startswith("@HD") and not parts[1].startswith("@") and not parts[2].startswith("@")
    assert all(p.count("\n") > n // 8 for p in parts), "every shard maps its share (boundaries fall on whole sub-ranges of the record index)"
    cat = str(tmp_path / "cat.sam")
    open(cat, "w").write("".join(parts))
    multi = str(tmp_path / "multi.sam")
    if not paired:   # a cold index cache: ONE process builds and writes it before the shard processes start, complete files only
        import glob
        for fn in glob.glob(fa + "-*.ngm"):
            os.remove(fn)
    c = subprocess.run([CLI, "-r", fa, "-o", multi, "--batch-size", "1024", "-g", "0,0", "--shard-output"] + inp, capture_output=True, text=True)
    assert c.returncode == 0, c.stderr[-2000:]
    assert "2 shards appended" in c.stderr and not os.path.exists(multi + ".shard1")
    if not paired:
        assert "building it once" in c.stderr and c.stderr.count("Reading reference index from") == 2, c.stderr[-3000:]
        assert not glob.glob(fa + "-*.tmp.*") and len(glob.glob(fa + "-*.ngm")) == 2
    a, b, m = _body(one), _body(cat), _body(multi)
    assert len(a) == len(b) == len(m) and len([l for l in a if not l.startswith("@")]) == n
    # the product's one reduction -- every shard hands its int64[8] statistics to the parent, which prints ONE summed line
    # (on distinct GPUs the shards also run the ncclAllReduce among themselves; two shards of one GPU cannot: RCCL refuses the duplicate)
    import re
    done = re.findall(r"Done \((\d+) reads mapped \([0-9.]+%\), (\d+) reads not mapped, (\d+) lines written\)", c.stderr)
    summed = re.search(r"Done, 2 shards summed \((\d+) reads mapped \([0-9.]+%\), (\d+) reads not mapped, (\d+) lines written; (\d+) reads; (\d+) pairs with both mates mapped, (\d+) of them broken", c.stderr)
    assert len(done) == 2 and summed, c.stderr[-3000:]
    assert [int(x) for x in summed.groups()[:3]] == [sum(int(d[k]) for d in done) for k in range(3)]
    assert int(summed.group(4)) == n and int(summed.group(3)) == len([l for l in m if not l.startswith("@")])
    if paired:
        recs = [l.split("\t") for l in m if not l.startswith("@")]
        mapped_both = sum(1 for f in recs if (int(f[1]) & 0x40) and not (int(f[1]) & 0x4) and not (int(f[1]) & 0x8))
        assert abs(int(summed.group(5)) - mapped_both) <= n // 100, (summed.group(5), mapped_both)   # (the writer's identity filter can still unmap a mate of a counted pair)
    else:
        assert int(summed.group(5)) == 0
    if not paired:
        assert a == b and a == m
    else:
        for other in (b, m):
            diff = [(x, y) for x, y in zip(a, other) if x != y]
            assert len(diff) <= n // 200, diff[:2]
            assert all(x.split("\t")[0] == y.split("\t")[0] for x, y in diff)


def _bs_reads(contigs, n, paired, seed):
    """Bisulfite-converted reads of a directional library: a first mate (or single read) shows most unmethylated C of the strand it
    was sequenced from as T, a second mate -- the reverse complement of that strand -- shows G as A, whichever strand of the
    genome the fragment came from (that is what CS::RunBatch's mutateFrom / mutateTo assume, src/CS.cpp:356-376)."""
    rng = np.random.default_rng(seed)

    def convert(reads, second):
        out = []
        frm, to = (ord("G"), ord("A")) if second else (ord("C"), ord("T"))
        for name, seq, qual in reads:
            s = seq.copy()
            m = (s == frm) & (rng.random(len(s)) < 0.9)
            s[m] = to
            out.append((name, s, qual))
        return out
    if paired:
        r1, r2 = S.make_reads(contigs, n, 100, seed=seed, sub_rate=0.01, indel_rate=0.002, paired=True)
        return convert(r1, False), convert(r2, True)
    return convert(S.make_reads(contigs, n, 100, seed=seed, sub_rate=0.01, indel_rate=0.002), False), None


@needs
@pytest.mark.parametrize("layout", ["single-end", "paired-end", "single-end-top3"])
def test_bisulfite_mapping_real_program_with_plugin_vs_ngm_hip(tmp_path, layout):
    """`--bs-mapping`.  The stock program cannot run this mode here (it excludes --affine, and the OpenCL
    backend has no device), so the oracle is the REAL program with this library behind IAlignment: its own CS::PrefixMutateSearch
    (src/CS.cpp:54-112), skip-0 index, ScoreBuffer direction bytes, computeCigarMD callers and SAMWriter (ZS tag) drive BatchScore /
    BatchAlign -- against ngm-hip, whose k-mer mutation search, direction bits and ZS tag are this repository's own."""
    contigs = S.make_genome([300000, 200001], seed=911, repeat_families=6, repeat_len=400, copies=4)
    fa = str(tmp_path / "ref.fa")
    with open(fa, "wb") as f:
        for i, g in enumerate(contigs):
            f.write(b">chr%d\n" % (i + 1))
            b = g.tobytes()
            for o in range(0, len(b), 70):
                f.write(b[o:o + 70] + b"\n")
    paired = layout == "paired-end"
    topn = ["-n", "3"] if layout == "single-end-top3" else []   # round 5: `--bs-mapping -n N` (the reference only excludes --affine, --slam-seq and --end-to-end: Config.cpp:448-470; ScoreBuffer::topNSE)
    r1, r2 = _bs_reads(contigs, 1500 if paired else 2500, paired, 912)
    fq = str(tmp_path / "reads.fq")
    if paired:
        S.write_fastq(fq, [x for pair in zip(r1, r2) for x in pair])
        inp, n = ["-p", "-q", fq], 2 * len(r1)
    else:
        S.write_fastq(fq, r1)
        inp, n = ["-q", fq], len(r1)
    d1 = tmp_path / "plug"
    d1.mkdir()
    fa1 = str(d1 / "ref.fa")
    os.link(fa, fa1)
    plug, ours = str(d1 / "plugin.sam"), str(tmp_path / "ours.sam")
    log = _run(DROPIN, fa1, inp + ["--bs-mapping"] + topn, plug, str(d1))
    c = subprocess.run([CLI, "-r", fa, "-o", ours, "--bs-mapping"] + topn + inp, capture_output=True, text=True)
    assert c.returncode == 0, c.stderr[-2000:]
    if topn:   # several records per read: compared as lists, in the order written per read
        def recs(p):
            d = {}
            for l in open(p):
                if not l.startswith("@"):
                    d.setdefault(l.split("\t", 1)[0], []).append(l)
            return d
        a, b = recs(plug), recs(ours)
        assert set(a) == set(b) and len(a) == n and sum(len(v) for v in a.values()) > n, "some reads must have several alignments"
        diff = [(a[k], b[k]) for k in a if a[k] != b[k]]
        print("reads differing:", len(diff), "of", len(a))
        assert not diff, str(diff[:1])[:1500]
        return
    rec = lambda p: {(l.split("\t", 2)[0], int(l.split("\t", 2)[1]) & 0xC0): l for l in open(p) if not l.startswith("@")}
    a, b = rec(plug), rec(ours)
    assert set(a) == set(b) and len(a) == n
    mapped = sum(1 for v in a.values() if not int(v.split("\t")[1]) & 4)
    assert mapped > 0.8 * n, "the converted reads must map (%d of %d)" % (mapped, n)
    assert all("\tZS:Z:" in v for v in a.values() if not int(v.split("\t")[1]) & 4)
    diff = [(a[k], b[k]) for k in a if a[k] != b[k]]
    print("records differing:", len(diff), "of", len(a))
    assert not diff, str(diff[:2])[:1500]


@needs
@pytest.mark.parametrize("layout", ["single-end", "paired-end"])
@pytest.mark.parametrize("slam", [1, 2, 4, 5, 6, 7])
def test_slam_seq_real_program_with_plugin_vs_ngm_hip(tmp_path, layout, slam):
    """`--slam-seq <n>`: 1 = conversion-aware NM / identity plus the TC / RA / MP tags, 2 = the strand-specific
    SLAM-seq score tables as well, 4 = the WEIGHTED candidate search (src/CS.cpp:57-92: every read k-mer and its single C > T
    conversions, float votes of 1 / (convertible bases + 1) -- here csrc/cs_slam_device.h, there the reference's own CS on the host:
    identical SAM means identical candidate sets, float maxima (XE:i) and candidate order).  As for bisulfite the oracle is the REAL program with this library behind IAlignment: its
    ScoreBuffer / AlignmentBuffer direction bytes and SAMWriter::computeSlaSeqTags (src/writer/GenericReadWriter.h:87-187) over the
    adapter's Align::ExtendedData records -- against ngm-hip, which builds the three tags on the GPU (csrc/sam_device.h)."""
    contigs = S.make_genome([300000, 200001], seed=921, repeat_families=6, repeat_len=400, copies=4)
    fa = str(tmp_path / "ref.fa")
    with open(fa, "wb") as f:
        for i, g in enumerate(contigs):
            f.write(b">chr%d\n" % (i + 1))
            b = g.tobytes()
            for o in range(0, len(b), 70):
                f.write(b[o:o + 70] + b"\n")
    paired = layout == "paired-end"
    rng = np.random.default_rng(922)

    def convert(reads, second):   # 4-thiouridine: T read as C on the sequenced strand (second mates: A as G)
        out = []
        frm, to = (ord("A"), ord("G")) if second else (ord("T"), ord("C"))
        for name, seq, qual in reads:
            s = seq.copy()
            s[(s == frm) & (rng.random(len(s)) < 0.08)] = to
            out.append((name, s, qual))
        return out
    fq = str(tmp_path / "reads.fq")
    if paired:
        r1, r2 = S.make_reads(contigs, 1500, 100, seed=923, sub_rate=0.01, indel_rate=0.002, paired=True)
        r1, r2 = convert(r1, False), convert(r2, True)
        S.write_fastq(fq, [x for pair in zip(r1, r2) for x in pair])
        inp, n = ["-p", "-q", fq], 2 * len(r1)
    else:
        r1 = convert(S.make_reads(contigs, 2500, 100, seed=923, sub_rate=0.01, indel_rate=0.002), False)
        S.write_fastq(fq, r1)
        inp, n = ["-q", fq], len(r1)
    d1 = tmp_path / "plug"
    d1.mkdir()
    fa1 = str(d1 / "ref.fa")
    os.link(fa, fa1)
    plug, ours = str(d1 / "plugin.sam"), str(tmp_path / "ours.sam")
    _run(DROPIN, fa1, inp + ["--slam-seq", str(slam)], plug, str(d1))
    c = subprocess.run([CLI, "-r", fa, "-o", ours, "--slam-seq", str(slam)] + inp, capture_output=True, text=True)
    assert c.returncode == 0, c.stderr[-2000:]
    rec = lambda p: {(l.split("\t", 2)[0], int(l.split("\t", 2)[1]) & 0xC0): l for l in open(p) if not l.startswith("@")}
    a, b = rec(plug), rec(ours)
    assert set(a) == set(b) and len(a) == n
    mapped = [v for v in a.values() if not int(v.split("\t")[1]) & 4]
    assert len(mapped) > 0.9 * n
    assert all("\tTC:i:" in v and "\tRA:Z:" in v for v in mapped)
    assert sum(int(v.split("\tTC:i:")[1].split("\t")[0]) for v in mapped) > n, "the conversions must be counted"
    diff = [(a[k], b[k]) for k in a if a[k] != b[k]]
    print("records differing:", len(diff), "of", len(a))
    assert not diff, str(diff[:2])[:1500]
    if slam in (1, 6):
        # ... with -n 3 (several alignments per read, each with its tags; single-end as in the reference) and as BAM (TC / RA / MP in the tag block)
        if not paired:
            plug3, ours3 = str(d1 / "plugin3.sam"), str(tmp_path / "ours3.sam")
            _run(DROPIN, fa1, inp + ["--slam-seq", str(slam), "-n", "3"], plug3, str(d1))
            c = subprocess.run([CLI, "-r", fa, "-o", ours3, "--slam-seq", str(slam), "-n", "3"] + inp, capture_output=True, text=True)
            assert c.returncode == 0, c.stderr[-2000:]
            body = lambda p: sorted(l for l in open(p) if not l.startswith("@"))
            assert body(plug3) == body(ours3)
        from test_gpu_bam import decode_bam
        plugb, oursb = str(d1 / "plugin.bam"), str(tmp_path / "ours.bam")
        _run(DROPIN, fa1, inp + ["--slam-seq", str(slam), "--bam"], plugb, str(d1))
        c = subprocess.run([CLI, "-r", fa, "-o", oursb, "--slam-seq", str(slam), "--bam"] + inp, capture_output=True, text=True)
        assert c.returncode == 0, c.stderr[-2000:]
        ta, ra, xa = decode_bam(plugb)
        tb, rb, xb = decode_bam(oursb)
        assert ra == rb and len(xa) == len(xb) == n
        bad = [(x, y) for x, y in zip(xa, xb) if x != y]
        assert not bad, str(bad[:1])[:1500]
        assert any(b"TCi" in x["tags"] or b"TCC" in x["tags"] or b"TCc" in x["tags"] for x in xa), "the BAM records carry the SLAM-seq tags"
    # the host formatter (NGM_HIP_HOST_SAM=1) writes the same tags
    host = str(tmp_path / "host.sam")
    c = subprocess.run([CLI, "-r", fa, "-o", host, "--slam-seq", str(slam)] + inp, capture_output=True, text=True, env=dict(os.environ, NGM_HIP_HOST_SAM="1"))
    assert c.returncode == 0, c.stderr[-2000:]
    strip = lambda p: [l for l in open(p) if not l.startswith("@PG")]
    assert strip(host) == strip(ours)


def test_one_shard_process_runs_the_rccl_all_reduce(tmp_path):
    """The collective itself: with NGM_HIP_SHARD_SINGLE=1 `-g 0 --shard-output` goes through the shard machinery with ONE shard process --
    a communicator of one rank (what a 1-GPU box allows: RCCL refuses two ranks on one GPU) -- and that process runs ncclAllReduce on its
    int64[8] statistics (librccl loaded at run time, the id made by the parent).  The reduced vector equals the parent's sum."""
    import re
    fa, inp, n = _case(tmp_path, True)
    out = str(tmp_path / "one_shard.sam")
    env = dict(os.environ, NGM_HIP_SHARD_SINGLE="1")
    c = subprocess.run([CLI, "-r", fa, "-o", out, "-g", "0", "--shard-output"] + inp, capture_output=True, text=True, env=env)
    assert c.returncode == 0, c.stderr[-2000:]
    ar = re.search(r"Statistics all-reduce over 1 GPUs \(RCCL, \d+ us\): (\d+) reads, (\d+) mapped, (\d+) not mapped, (\d+) lines written; (\d+) pairs with both mates mapped, (\d+) of them broken", c.stderr)
    summed = re.search(r"Done, 1 shards summed \((\d+) reads mapped \([0-9.]+%\), (\d+) reads not mapped, (\d+) lines written; (\d+) reads; (\d+) pairs with both mates mapped, (\d+) of them broken", c.stderr)
    assert ar and summed, c.stderr[-3000:]
    assert (ar.group(1), ar.group(2), ar.group(3), ar.group(4), ar.group(5), ar.group(6)) == (summed.group(4), summed.group(1), summed.group(2), summed.group(3), summed.group(5), summed.group(6))
    assert int(ar.group(1)) == n
