"""-m gpu: parity on a GRCh38-LIKE k-mer spectrum ("what's weak" 1 / "next" 1).

Every other pipeline test maps against near-Poisson genomes (uniform random + a few small repeat families); there the paths that
handle long index lists, table / queue overflow, thousands of hits per read and hundreds of candidates serve 0.04 % of the reads.
tests/humanlike.py builds a genome with a heavy-tailed spectrum (one SINE-like family at 100 000 copies, LINE-like families,
satellite arrays, microsatellites, segmental duplications, isochores) on which `max_kfreq`'s automatic rule
(src/PrefixTable.cpp:150-194) and the "9 901 occurrences => unused" byte (:468-478) both fire, and draws half of the reads FROM the
repeats.  `ngm-hip --affine` must equal `ngm-core --affine -t 1` in every SAM field; the index files both programs write must be
identical; and no read may have lost its reference candidate order silently (`Candidate order replay` line of the log)."""
import filecmp
import os
import re
import subprocess

import pytest

import humanlike as H
import ref_files as RF
from test_gpu_cli import _sam
from test_gpu_cli import _sam_pe

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "nextgenmap_amd", "ngm-hip")
needs_ref = pytest.mark.skipif(not RF.have_reference_binary(), reason="reference binary not built (oracle/ngm_ref.mk)")

GENOME_BP = int(os.environ.get("NGM_TEST_HUMANLIKE_BP", 120_000_000))
N_SE = int(os.environ.get("NGM_TEST_HUMANLIKE_SE", 50_000))
N_PE = int(os.environ.get("NGM_TEST_HUMANLIKE_PE", 50_000))     # pairs
SINE_COPIES = int(os.environ.get("NGM_TEST_HUMANLIKE_SINE", 100_000))   # one ~300 bp family, 25 % of the bases


@pytest.fixture(scope="module")
def world(tmp_path_factory):
    from nextgenmap_amd import build
    build.build()
    d = tmp_path_factory.mktemp("humanlike")
    G = H.make_genome(total_bp=GENOME_BP, seed=7, sine_copies=SINE_COPIES)
    fa = str(d / "ref.fa")
    H.write_fasta(fa, G)
    refdir = d / "refrun"
    refdir.mkdir()
    os.link(fa, str(refdir / "ref.fa"))
    se = H.make_reads(G, N_SE, 150, seed=11)
    H.write_fastq(str(d / "se.fq"), se)
    r1, r2 = H.make_reads(G, N_PE, 150, seed=12, paired=True)
    H.write_fastq(str(d / "pe_1.fq"), r1)
    H.write_fastq(str(d / "pe_2.fq"), r2)
    return dict(dir=d, fa=fa, ref_fa=str(refdir / "ref.fa"), refdir=refdir, G=G)


def _both(world, tag, args, hip_extra=()):
    d = world["dir"]
    ref_sam, hip_sam = str(world["refdir"] / (tag + ".sam")), str(d / (tag + "_hip.sam"))
    r = RF.run_ngm(["-r", world["ref_fa"], "-o", ref_sam, "--affine", "-t", "1", "--no-progress"] + args, cwd=str(world["refdir"]), timeout=3000)
    log_ref = r.stdout + r.stderr
    assert "Done" in log_ref, log_ref[-1500:]
    c = subprocess.run([CLI, "-r", world["fa"], "-o", hip_sam, "--affine"] + args + list(hip_extra), capture_output=True, text=True)
    assert c.returncode == 0, c.stderr[-2000:]
    return ref_sam, hip_sam, log_ref, c.stderr


def _report(tag, a, b, log_hip):
    diff = [n for n in a if a[n] != b[n]]
    kinds = {}
    for n in diff:
        kind = (n[0] if isinstance(n, tuple) else n).split("_")[-1].split("/")[0]
        kinds[kind] = kinds.get(kind, 0) + 1
    print("%s: %d records, %d differ %s" % (tag, len(a), len(diff), kinds))
    for n in diff[:8]:
        x, y = a[n], b[n]
        print("  ", n, {k: (x[k], y[k]) for k in x if x[k] != y[k] and k != "tags"}, {k: (x["tags"].get(k), y["tags"].get(k)) for k in set(x["tags"]) | set(y["tags"]) if x["tags"].get(k) != y["tags"].get(k)})
    for line in log_hip.splitlines():
        if "Candidate search" in line or "Candidate order" in line or "candidates per read" in line:
            print("  ", line)
    return diff


@needs_ref
def test_index_files_identical_on_a_heavy_tailed_genome(world):
    """first run of each program builds and saves the index: same bytes (the usage rule, max_kfreq's automatic value)"""
    ref_sam, hip_sam, log_ref, log_hip = _both(world, "tiny", ["-q", str(world["dir"] / "se.fq")])
    kf_ref = re.search(r"Max. k-mer frequency set so (\d+)!", log_ref)
    kf_hip = re.search(r"max. k-mer frequency (\d+)", log_hip)
    assert kf_ref and kf_hip and kf_ref.group(1) == kf_hip.group(1), (kf_ref, kf_hip)
    assert int(kf_ref.group(1)) > 100, "the genome should make max_kfreq's automatic rule fire"
    ign = re.search(r"Number of repetitive k-mers ignored: (\d+)", log_ref)
    assert ign and int(ign.group(1)) > 0
    for suffix in ("-enc.2.ngm", "-ht-13-2.3.ngm"):
        assert filecmp.cmp(world["ref_fa"] + suffix, world["fa"] + suffix, shallow=False), suffix
    # and since that run mapped the single-end reads: compare them here
    a, b = _sam(ref_sam), _sam(hip_sam)
    assert set(a) == set(b) and len(a) == N_SE
    for pat in (r"Average read length: (\d+) \(min: (\d+), max: (\d+)\)", r"Corridor width: (\d+)", r"Estimated sensitivity: ([0-9.]+)"):
        assert re.search(pat, log_ref).groups() == re.search(pat, log_hip).groups(), pat
    diff = _report("single-end", a, b, log_hip)
    gave_up = re.search(r"Candidate order replay: (\d+) reads, (\d+) of them beyond", log_hip)
    assert gave_up, log_hip[-3000:]
    unknown = re.search(r"order left undetermined for (\d+) reads", log_hip)
    assert unknown and int(unknown.group(1)) == 0, log_hip[-3000:]
    assert len(diff) == 0, len(diff)


@needs_ref
def test_paired_end_on_a_heavy_tailed_genome(world):
    d = world["dir"]
    ref_sam, hip_sam, log_ref, log_hip = _both(world, "pe", ["-1", str(d / "pe_1.fq"), "-2", str(d / "pe_2.fq")])
    a, b = _sam_pe(ref_sam), _sam_pe(hip_sam)
    assert set(a) == set(b) and len(a) == 2 * N_PE
    diff = _report("paired-end", a, b, log_hip)
    unknown = re.search(r"order left undetermined for (\d+) reads", log_hip)
    assert unknown and int(unknown.group(1)) == 0, log_hip[-3000:]
    assert len(diff) == 0, len(diff)


# The limits below make a 120 Mbp genome behave like GRCh38 does with the product's real ones ("what's missing" 1): tables of
# cs_heavy2_kernel a 64th of their size, so that table passes overflow and start over with twice the parts, reads fail every class
# and reach the exact kernels -- the LDS table and, beyond its 10 813 hits, cs_global_kernel with tables from a pool that must grow --;
# 64 buckets per read in the bucket replay, so that every bucket is a window of its own (> 64 hits).
FORCED = "heavy_log2c=9,heavy_log2s=7,heavy_max0=1500,heavy_scratch=4096,gtable_pool_log2=12,order_buckets_log2=6"


def _counters(log):
    pat = {"heavy": r"heavy-read kernel for (\d+) reads", "exact_lds": r"table in LDS for (\d+)", "exact_global": r"in global memory for (\d+)",
           "second": r"Heavy-read kernel: (\d+) second passes", "restart": r"(\d+) table passes started over", "sent_on": r"(\d+) reads sent on",
           "regrown": r"table pool regrown (\d+) times", "replayed": r"Candidate order replay: (\d+) reads", "beyond": r"(\d+) of them beyond the LDS replay",
           "unknown": r"order left undetermined for (\d+) reads"}
    out = {}
    for k, p in pat.items():
        mm = re.search(p, log)
        assert mm, (k, log[-3000:])
        out[k] = int(mm.group(1))
    return out


@needs_ref
def test_grch38_sized_paths_under_forced_limits(world):
    """The paths only a GRCh38-sized index reaches with the real limits -- table passes started over (csrc/cs_heavy_device.h), the exact
    kernels for ordinary reads incl. cs_global_kernel and its pool (csrc/mapper_search.cpp), crowded buckets in the order replay
    (csrc/cs_order_bucket_device.h) -- forced on this genome by NGM_HIP_TEST_LIMITS, counted, and 100 000 paired-end records compared
    with `ngm-core --affine -t 1` (the reference's own overflow handling: src/CS.cpp:397-430)."""
    d = world["dir"]
    args = ["-1", str(d / "pe_1.fq"), "-2", str(d / "pe_2.fq")]
    ref_sam = str(world["refdir"] / "pe.sam")
    if not os.path.exists(ref_sam):   # (test_paired_end_on_a_heavy_tailed_genome leaves it; run alone: make it)
        r = RF.run_ngm(["-r", world["ref_fa"], "-o", ref_sam, "--affine", "-t", "1", "--no-progress"] + args, cwd=str(world["refdir"]), timeout=3000)
        assert "Done" in r.stdout + r.stderr
    hip_sam = str(d / "forced_hip.sam")
    c = subprocess.run([CLI, "-r", world["fa"], "-o", hip_sam, "--affine"] + args, capture_output=True, text=True, env=dict(os.environ, NGM_HIP_TEST_LIMITS=FORCED))
    assert c.returncode == 0, c.stderr[-2000:]
    k = _counters(c.stderr)
    print("forced limits:", k)
    assert k["heavy"] > 0 and k["second"] > 0 and k["restart"] > 0 and k["sent_on"] > 0, k
    assert k["exact_lds"] > 0 and k["exact_global"] > 0 and k["regrown"] > 0, k
    assert k["beyond"] > 0 and k["unknown"] == 0, k
    a, b = _sam_pe(ref_sam), _sam_pe(hip_sam)
    assert set(a) == set(b) and len(a) == 2 * N_PE and len(a) >= 100_000
    diff = _report("paired-end, forced limits", a, b, c.stderr)
    assert len(diff) == 0, len(diff)
    # the same reads with the product's own limits take none of those paths on this genome: that is why the test forces them
    base = subprocess.run([CLI, "-r", world["fa"], "-o", str(d / "unforced_hip.sam"), "--affine"] + args, capture_output=True, text=True)
    assert base.returncode == 0
    k0 = _counters(base.stderr)
    print("product limits:", k0)
    assert [l for l in open(hip_sam, "rb") if not l.startswith(b"@PG")] == [l for l in open(str(d / "unforced_hip.sam"), "rb") if not l.startswith(b"@PG")]


def test_order_replay_without_room_degrades_and_says_so(world):
    """a replay whose scratch cannot hold a read leaves that read's candidate order UNDETERMINED -- counted and printed, ties then
    resolve by position -- instead of ending the run.  A 64 KB pool (NGM_HIP_TEST_LIMITS) is too small for the reads of the repeat families."""
    d = world["dir"]
    out = str(d / "degraded.sam")
    c = subprocess.run([CLI, "-r", world["fa"], "-o", out, "--affine", "-q", str(d / "se.fq")], capture_output=True, text=True,
                       env=dict(os.environ, NGM_HIP_TEST_LIMITS="order_pool_kb=64"))
    assert c.returncode == 0, c.stderr[-2000:]
    k = _counters(c.stderr)
    print("replay pool of 64 KB:", k)
    assert k["beyond"] > 0 and k["unknown"] > 0, k
    assert sum(1 for l in open(out) if not l.startswith("@")) == N_SE


def test_three_order_replays_agree_on_a_heavy_tailed_genome(world):
    """The candidate order of a read beyond the LDS replay comes from cs_order_bucket_kernel (hits dealt into buckets); what it leaves, and
    everything with NGM_HIP_ORDER_NO_BUCKETS, from cs_order_kernel<true> (a table in global memory); NGM_HIP_ORDER_LDS_BIG keeps the reads
    of up to 49 152 hits with the LDS replay (its time line in a slice of global memory).  Three implementations of CS::AddLocationStd's
    rList order (src/CS.cpp:196-211): byte-identical SAM."""
    d = world["dir"]
    args = ["-1", str(d / "pe_1.fq"), "-2", str(d / "pe_2.fq")]

    def run(tag, env):
        out = str(d / ("order_" + tag + ".sam"))
        c = subprocess.run([CLI, "-r", world["fa"], "-o", out, "--affine"] + args, capture_output=True, text=True, env=dict(os.environ, **env))
        assert c.returncode == 0, c.stderr[-2000:]
        return [l for l in open(out, "rb") if not l.startswith(b"@PG")], c.stderr
    base, log = run("buckets", {})
    beyond = re.search(r"Candidate order replay: (\d+) reads, (\d+) of them beyond", log)
    assert beyond and int(beyond.group(2)) > 0, log[-2000:]
    unknown = re.search(r"order left undetermined for (\d+) reads", log)
    assert unknown and int(unknown.group(1)) == 0, log[-2000:]
    with_table = re.search(r"(\d+) of them with a table there", log)
    assert with_table and int(with_table.group(1)) < int(beyond.group(2)), log[-2000:]
    table, log1 = run("table", {"NGM_HIP_ORDER_NO_BUCKETS": "1"})
    beyond1 = re.search(r"Candidate order replay: (\d+) reads, (\d+) of them beyond", log1)   # (fewer: this run keeps the LDS replay's global time line)
    assert beyond1 and int(beyond1.group(2)) > 0 and re.search(r"(\d+) of them with a table there", log1).group(1) == beyond1.group(2)
    assert table == base
    lds_big, log2 = run("lds-big", {"NGM_HIP_ORDER_LDS_BIG": "1"})
    beyond2 = re.search(r"Candidate order replay: (\d+) reads, (\d+) of them beyond", log2)
    assert beyond2 and 0 < int(beyond2.group(2)) <= int(beyond.group(2))
    assert lds_big == base


def test_single_end_linear_personality_through_the_drop_in(world):
    """the DEFAULT (linear-gap) personality on the same reads: the reference's own program with this library behind IAlignment
    (oracle/build_dropin.sh) against ngm-hip"""
    dropin = os.path.join(ROOT, "oracle", "_ref", "dropin", "ngm-core-hip")
    if not os.path.exists(dropin):
        pytest.skip("drop-in build not present (oracle/build_dropin.sh)")
    d = world["dir"]
    n = min(N_SE, 20_000) * 4
    fq = str(d / "se_head.fq")
    with open(str(d / "se.fq"), "rb") as f, open(fq, "wb") as g:
        for i, line in enumerate(f):
            if i >= n:
                break
            g.write(line)
    ref_sam, hip_sam = str(world["refdir"] / "lin.sam"), str(d / "lin_hip.sam")
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "nextgenmap_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([dropin, "-r", world["ref_fa"], "-q", fq, "-o", ref_sam, "-t", "1", "--no-progress"], capture_output=True, text=True, cwd=str(world["refdir"]), env=env, timeout=3000)
    assert "Done" in r.stdout + r.stderr, (r.stdout + r.stderr)[-1500:]
    c = subprocess.run([CLI, "-r", world["fa"], "-q", fq, "-o", hip_sam], capture_output=True, text=True)
    assert c.returncode == 0, c.stderr[-2000:]
    a, b = _sam(ref_sam), _sam(hip_sam)
    assert set(a) == set(b) and len(a) == n // 4
    diff = _report("linear personality (drop-in)", a, b, c.stderr)
    assert len(diff) == 0, len(diff)


def test_weighted_slam_seq_search_on_a_heavy_tailed_genome(world):
    """`--slam-seq 4` (float votes in the reference's order, csrc/cs_slam_device.h) where reads carry tens of thousands of hits and the
    persistent workgroups' slices are too small for some of them: the reference's own candidate search on the host (the drop-in)
    against ngm-hip, 4 000 reads with T > C conversions."""
    import numpy as np
    dropin = os.path.join(ROOT, "oracle", "_ref", "dropin", "ngm-core-hip")
    if not os.path.exists(dropin):
        pytest.skip("drop-in build not present (oracle/build_dropin.sh)")
    d = world["dir"]
    rng = np.random.default_rng(5)
    fq = str(d / "slam.fq")
    with open(str(d / "se.fq"), "rb") as f, open(fq, "wb") as g:
        for i in range(4000):
            name, seq, plus, qual = f.readline(), bytearray(f.readline()), f.readline(), f.readline()
            for j in range(len(seq) - 1):
                if seq[j] == ord("T") and rng.random() < 0.06:
                    seq[j] = ord("C")
            g.write(name + bytes(seq) + plus + qual)
    ref_sam, hip_sam = str(world["refdir"] / "slam.sam"), str(d / "slam_hip.sam")
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "nextgenmap_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([dropin, "-r", world["ref_fa"], "-q", fq, "-o", ref_sam, "-t", "1", "--no-progress", "--slam-seq", "4"], capture_output=True, text=True,
                       cwd=str(world["refdir"]), env=env, timeout=3000)
    assert "Done" in r.stdout + r.stderr, (r.stdout + r.stderr)[-1500:]
    c = subprocess.run([CLI, "-r", world["fa"], "-q", fq, "-o", hip_sam, "--slam-seq", "4"], capture_output=True, text=True)
    assert c.returncode == 0, c.stderr[-2000:]
    a, b = _sam(ref_sam), _sam(hip_sam)
    assert set(a) == set(b) and len(a) == 4000
    diff = _report("weighted SLAM-seq search (drop-in)", a, b, c.stderr)
    assert len(diff) == 0, len(diff)


def _head(src, dst, records):
    with open(src, "rb") as f, open(dst, "wb") as g:
        for i, line in enumerate(f):
            if i >= 4 * records:
                break
            g.write(line)


def _sam_multi(path):
    """name -> sorted list of whole records (reads with several alignments under -n)"""
    recs = {}
    for line in open(path):
        if not line.startswith("@"):
            recs.setdefault(line.split("\t", 1)[0], []).append(line)
    return {k: sorted(v) for k, v in recs.items()}


@needs_ref
@pytest.mark.parametrize("mode", ["topn3", "topn2-strata", "end-to-end", "fast-pairing", "pe-end-to-end", "pe-strata", "250bp-sensitive", "75bp"])
def test_other_selection_modes_on_a_heavy_tailed_genome(world, mode):
    """the selection modes whose outcome hangs on the reference's candidate ORDER among equal scores (-n cuts a sorted list, --strata
    counts the equally best, top1SE keeps the first) where equal scores are the rule: reads from repeat families with hundreds of
    candidates.  12 000 reads (6 000 pairs for --fast-pairing), every SAM line equal to `ngm-core --affine -t 1`'s."""
    d = world["dir"]
    if mode in ("fast-pairing", "pe-end-to-end", "pe-strata"):
        f1, f2 = str(d / "fp_1.fq"), str(d / "fp_2.fq")
        _head(str(d / "pe_1.fq"), f1, 6000)
        _head(str(d / "pe_2.fq"), f2, 6000)
        args, n = ["-1", f1, "-2", f2] + {"fast-pairing": ["--fast-pairing"], "pe-end-to-end": ["-e"], "pe-strata": ["--strata"]}[mode], 12000
    elif mode in ("250bp-sensitive", "75bp"):
        # other read lengths: 250 bp at 8 % substitutions with the wide band of BASELINE config 5 (-C 40 --sensitive), and 75 bp (few k-mers)
        rl = 250 if mode.startswith("250") else 75
        fq = str(d / ("len%d.fq" % rl))
        n = 1200 if rl == 250 else 5000   # (the reference needs 50 ms per 250 bp read here: ~100 candidates each, 81-column band, one thread)
        H.write_fastq(fq, H.make_reads(world["G"], n, rl, seed=31 + rl, sub_rate=0.08 if rl == 250 else 0.01, indel_rate=0.01 if rl == 250 else 0.001))
        args = ["-q", fq] + (["-C", "40", "--sensitive"] if rl == 250 else [])
    else:
        fq = str(d / "modes.fq")
        _head(str(d / "se.fq"), fq, 12000)
        args = ["-q", fq] + {"topn3": ["-n", "3"], "topn2-strata": ["-n", "2", "--strata"], "end-to-end": ["-e"]}[mode]
        n = 12000
    ref_sam, hip_sam, log_ref, log_hip = _both(world, "mode_" + mode, args)
    a, b = _sam_multi(ref_sam), _sam_multi(hip_sam)
    assert set(a) == set(b) and sum(len(v) for v in a.values()) >= (n if "strata" not in mode else 1)
    diff = [k for k in a if a[k] != b[k]]
    print("%s: %d reads, %d records, %d reads differ" % (mode, len(a), sum(len(v) for v in a.values()), len(diff)))
    for k in diff[:4]:
        print("  ", a[k][:2], b[k][:2])
    for line in log_hip.splitlines():
        if "Candidate order" in line:
            print("  ", line)
    assert not diff, len(diff)


@pytest.mark.parametrize("what", ["bs-mapping", "slam-seq-2"])
def test_converted_reads_on_a_heavy_tailed_genome(world, what):
    """`--bs-mapping` (every T > C variant of every read k-mer against a skip-0 index, the strand-specific score tables) and `--slam-seq 2`
    on the repeat-rich genome: the reference's own program with this library behind IAlignment against ngm-hip, 3 000 converted reads."""
    import numpy as np
    dropin = os.path.join(ROOT, "oracle", "_ref", "dropin", "ngm-core-hip")
    if not os.path.exists(dropin):
        pytest.skip("drop-in build not present (oracle/build_dropin.sh)")
    d = world["dir"]
    rng = np.random.default_rng(9)
    frm, to, rate = (ord("C"), ord("T"), 0.9) if what == "bs-mapping" else (ord("T"), ord("C"), 0.06)
    fq = str(d / (what + ".fq"))
    with open(str(d / "se.fq"), "rb") as f, open(fq, "wb") as g:
        for i in range(3000):
            name, seq, plus, qual = f.readline(), bytearray(f.readline()), f.readline(), f.readline()
            for j in range(len(seq) - 1):
                if seq[j] == frm and rng.random() < rate:
                    seq[j] = to
            g.write(name + bytes(seq) + plus + qual)
    opt = ["--bs-mapping"] if what == "bs-mapping" else ["--slam-seq", "2"]
    sub = world["refdir"] / what          # (a bisulfite run builds its own skip-0 index cache: keep the two programs' files apart)
    sub.mkdir(exist_ok=True)
    rfa = str(sub / "ref.fa")
    if not os.path.exists(rfa):
        os.link(world["fa"], rfa)
    hsub = d / ("hip_" + what)
    hsub.mkdir(exist_ok=True)
    hfa = str(hsub / "ref.fa")
    if not os.path.exists(hfa):
        os.link(world["fa"], hfa)
    ref_sam, hip_sam = str(sub / "out.sam"), str(d / (what + "_hip.sam"))
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "nextgenmap_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([dropin, "-r", rfa, "-q", fq, "-o", ref_sam, "-t", "1", "--no-progress"] + opt, capture_output=True, text=True, cwd=str(sub), env=env, timeout=3000)
    assert "Done" in r.stdout + r.stderr, (r.stdout + r.stderr)[-1500:]
    c = subprocess.run([CLI, "-r", hfa, "-q", fq, "-o", hip_sam] + opt, capture_output=True, text=True)
    assert c.returncode == 0, c.stderr[-2000:]
    a, b = _sam(ref_sam), _sam(hip_sam)
    assert set(a) == set(b) and len(a) == 3000
    assert sum(1 for n in a if not a[n]["flag"] & 4) > 2000
    diff = _report(what + " (drop-in)", a, b, c.stderr)
    assert len(diff) == 0, len(diff)
