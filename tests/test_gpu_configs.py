"""-m gpu: the BASELINE.json configurations that round 1 left without a pipeline-level parity test.

  config 5  250 bp single-end reads at 15 % divergence (12 % substitutions + 3 % indel bases), -C 40 (corridor 80, the wide
            band), --sensitive, on a repeat-rich genome: `ngm-hip --affine` == `ngm --affine -t 1`, every SAM field;
  config 3  150 bp paired-end through the command line (two files), same comparison;
  workers   the pipelined CLI gives byte-identical SAM whatever the number of workers / the batch size (the paired-end
            running mean insert size is shared state taken in batch order, include/ngm_pipeline.h: ngm_pair_state);
  big       tests/big_parity.py's 60 Mbp / 100 000 pairs + 200 000 single-end reads as a collected test (it takes a minute
            of GPU time; the reference side dominates)."""
import os
import subprocess

import numpy as np
import pytest

import ref_files as RF
import simulate as S
from test_gpu_cli import _sam, _sam_pe

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "nextgenmap_amd", "ngm-hip")
needs_ref = pytest.mark.skipif(not RF.have_reference_binary(), reason="reference binary not built (oracle/ngm_ref.mk)")


def _write_fasta(path, contigs):
    with open(path, "wb") as f:
        for i, g in enumerate(contigs):
            f.write(b">chr%d\n" % (i + 1))
            b = g.tobytes()
            for o in range(0, len(b), 60):
                f.write(b[o:o + 60] + b"\n")


def _run_both(tmp_path, fa, args, hip_extra=()):
    from nextgenmap_amd import build
    build.build()
    d1 = tmp_path / "refrun"
    d1.mkdir(exist_ok=True)
    fa1 = str(d1 / "ref.fa")
    if not os.path.exists(fa1):
        os.link(fa, fa1)
    r = RF.run_ngm(["-r", fa1, "-o", str(d1 / "out.sam"), "--affine", "-t", "1", "--no-progress"] + args, cwd=str(d1), timeout=3000)
    assert "Done" in (r.stdout + r.stderr), (r.stdout + r.stderr)[-1500:]
    c = subprocess.run([CLI, "-r", fa, "-o", str(tmp_path / "hip.sam"), "--affine"] + args + list(hip_extra), capture_output=True, text=True)
    assert c.returncode == 0, c.stderr[-2000:]
    return str(d1 / "out.sam"), str(tmp_path / "hip.sam"), r.stdout + r.stderr, c.stderr


@needs_ref
def test_config5_250bp_15pct_divergence_wide_band_sensitive(tmp_path):
    contigs = S.make_genome([3_000_000, 2_000_001], seed=501, repeat_families=40, repeat_len=800, copies=12, divergence=0.01)
    fa = str(tmp_path / "ref.fa")
    _write_fasta(fa, contigs)
    # 12 % substitutions; indel events of geometric length (mean 2, at most 5) at 1.5 % per base = 3 % of the bases in indels
    reads = S.make_reads(contigs, 6000, 250, seed=502, sub_rate=0.12, indel_rate=0.015, max_indel=5)
    fq = str(tmp_path / "reads.fq")
    S.write_fastq(fq, reads)
    ref_sam, hip_sam, log_ref, log_hip = _run_both(tmp_path, fa, ["-q", fq, "-C", "40", "--sensitive"])
    import re
    for pat in (r"Average read length: (\d+) \(min: (\d+), max: (\d+)\)", r"Corridor width: (\d+)", r"Estimated sensitivity: ([0-9.]+)"):
        assert re.search(pat, log_ref).groups() == re.search(pat, log_hip).groups(), pat
    assert re.search(r"Corridor width: (\d+)", log_hip).group(1) == "80"
    a, b = _sam(ref_sam), _sam(hip_sam)
    assert set(a) == set(b) and len(a) == 6000
    mapped = sum(1 for n in a if not a[n]["flag"] & 4)
    diff = [(n, a[n], b[n]) for n in a if a[n] != b[n]]
    print("config 5: %d of %d reads mapped by the reference; records differing: %d" % (mapped, len(a), len(diff)))
    for n, x, y in diff[:6]:
        print(n, {k: (x[k], y[k]) for k in x if x[k] != y[k]}, "NH", x["tags"].get("NH"), y["tags"].get("NH"), "XE", x["tags"].get("XE"))
    assert mapped > 0.5 * len(a)  # the case is hard but not degenerate
    assert len(diff) == 0, (len(diff), str(diff[:2])[:1500])


@needs_ref
def test_config3_150bp_paired_end_two_files(tmp_path):
    contigs = S.make_genome([3_000_000, 2_000_001], seed=511, repeat_families=40, repeat_len=800, copies=12, divergence=0.01)
    fa = str(tmp_path / "ref.fa")
    _write_fasta(fa, contigs)
    r1, r2 = S.make_reads(contigs, 15000, 150, seed=512, sub_rate=0.01, indel_rate=0.001, paired=True)
    # quality lines that look like header / separator lines: the parallel record index finds record starts inside byte ranges
    # ('@' line whose second-next line starts with '+': only headers, ngm_cli.cpp build_fastq_index) and must not be fooled
    def tricky(rs):
        return [(nm, sq, (b"@" if i % 3 == 0 else b"+" if i % 3 == 1 else ql[:1]) + ql[1:]) for i, (nm, sq, ql) in enumerate(rs)]
    r1, r2 = tricky(r1), tricky(r2)
    f1, f2 = str(tmp_path / "pe_1.fq"), str(tmp_path / "pe_2.fq")
    S.write_fastq(f1, r1)
    S.write_fastq(f2, r2)
    ref_sam, hip_sam, _, log_hip = _run_both(tmp_path, fa, ["-1", f1, "-2", f2])
    assert "memory-mapped plain FASTQ" in log_hip
    a, b = _sam_pe(ref_sam), _sam_pe(hip_sam)
    assert set(a) == set(b) and len(a) == 2 * len(r1)
    diff = [(n, a[n], b[n]) for n in a if a[n] != b[n]]
    print("config 3 shape: records differing:", len(diff), "of", len(a))
    assert len(diff) == 0, (len(diff), diff[:3])


def test_pipeline_output_independent_of_workers_and_batches(tmp_path):
    """Same input through: 1 worker / one big batch / serial reader  vs  3 workers / small batches / mapped-file reader (and gz
    input): byte-identical SAM bodies.  Repeat-rich genome, so that the paired-end tie-breaks (running mean) do occur."""
    from nextgenmap_amd import build
    build.build()
    contigs = S.make_genome([2_000_000, 1_000_001], seed=521, repeat_families=40, repeat_len=800, copies=12, divergence=0.01)
    fa = str(tmp_path / "ref.fa")
    _write_fasta(fa, contigs)
    r1, r2 = S.make_reads(contigs, 30000, 125, seed=522, sub_rate=0.015, indel_rate=0.002, paired=True)
    f1, f2 = str(tmp_path / "pe_1.fq"), str(tmp_path / "pe_2.fq")
    S.write_fastq(f1, r1)
    S.write_fastq(f2, r2)
    subprocess.check_call("gzip -k %s %s" % (f1, f2), shell=True)

    def run(tag, extra, inputs, env=None):
        out = str(tmp_path / (tag + ".sam"))
        c = subprocess.run([CLI, "-r", fa, "-o", out, "--affine"] + inputs + extra, capture_output=True, text=True, env=dict(os.environ, **env) if env else None)
        assert c.returncode == 0, c.stderr[-2000:]
        body = [l for l in open(out, "rb") if not l.startswith(b"@PG")]
        return body, c.stderr
    base, log0 = run("serial", ["--workers", "1", "--serial-reader", "--batch-size", "1000000"], ["-1", f1, "-2", f2])
    assert "serial reader" in log0
    piped, log1 = run("piped", ["--workers", "3", "--batch-size", "4096"], ["-1", f1, "-2", f2])
    assert "memory-mapped plain FASTQ" in log1
    # the walk along the reference's score buffer (the pairs NextGenMap loses) is sequential state like the running mean:
    # the same count whatever the workers / batches
    import re
    c0 = re.search(r"Pairs lost as NextGenMap loses them .*: (\d+)", log0)
    c1 = re.search(r"Pairs lost as NextGenMap loses them .*: (\d+)", log1)
    assert c0 and c1 and c0.groups() == c1.groups(), (log0[-600:], log1[-600:])
    assert len(base) == len(piped) and base == piped
    # the pairs with choices: settled by pair_choice_kernel (what a repeat-rich genome gets by itself) or walked by the host -- same records
    on_gpu, _ = run("pair-gpu", ["--workers", "2", "--batch-size", "8192"], ["-1", f1, "-2", f2], env={"NGM_HIP_GPU_PAIR_CHOICE": "1"})
    on_host, _ = run("pair-host", ["--workers", "2", "--batch-size", "8192"], ["-1", f1, "-2", f2], env={"NGM_HIP_HOST_PAIR_CHOICE": "1"})
    assert on_gpu == base and on_host == base
    gz, log2 = run("gz", ["--workers", "2", "--batch-size", "10000"], ["-1", f1 + ".gz", "-2", f2 + ".gz"])
    assert "gzip FASTQ inflated to memory" in log2
    assert gz == base
    gz1, log3 = run("gz-serial", ["--workers", "2", "--serial-reader"], ["-1", f1 + ".gz", "-2", f2 + ".gz"])
    assert "serial reader" in log3 and gz1 == base
    # single-end, odd batch size
    se0, _ = run("se0", ["--workers", "1", "--serial-reader"], ["-q", f1])
    se1, _ = run("se1", ["--workers", "3", "--batch-size", "5001"], ["-q", f1])
    assert se0 == se1


def test_kernel_variants_give_identical_sam(tmp_path):
    """The run-time selectable kernel variants are implementations of ONE result: candidate search on 1 / 2 / 3 (default) / 4
    waves per read, the 32-bit score / align kernels instead of the packed 16-bit ones, CIGAR / MD strings built on the host
    instead of the GPU, candidate slots through the region cursors only.  Paired-end on a repeat-rich genome (ties, repeat
    families: the reads that overflow the per-wave queues), both personalities: byte-identical SAM bodies."""
    from nextgenmap_amd import build
    build.build()
    contigs = S.make_genome([1_500_000, 900_001], seed=611, repeat_families=40, repeat_len=800, copies=14, divergence=0.01)
    fa = str(tmp_path / "ref.fa")
    _write_fasta(fa, contigs)
    r1, r2 = S.make_reads(contigs, 20000, 150, seed=612, sub_rate=0.015, indel_rate=0.002, paired=True)
    f1, f2 = str(tmp_path / "pe_1.fq"), str(tmp_path / "pe_2.fq")
    S.write_fastq(f1, r1)
    S.write_fastq(f2, r2)

    def run(tag, personality, env):
        out = str(tmp_path / (tag + ".sam"))
        e = dict(os.environ)
        e.update(env)
        c = subprocess.run([CLI, "-r", fa, "-o", out, "-1", f1, "-2", f2] + personality, capture_output=True, text=True, env=e)
        assert c.returncode == 0, c.stderr[-2000:]
        return [l for l in open(out, "rb") if not l.startswith(b"@PG")]
    for personality in (["--affine"], []):
        base = run("base", personality, {})
        assert len(base) > 40000
        for tag, env in (("w1", {"NGM_HIP_CS_WAVES": "1"}), ("w2", {"NGM_HIP_CS_WAVES": "2"}), ("w4", {"NGM_HIP_CS_WAVES": "4"}),
                         ("dp32", {"NGM_HIP_ALIGN_32BIT": "1", "NGM_HIP_SCORE_32BIT": "1"}), ("hostcigar", {"NGM_HIP_HOST_CIGAR": "1"}),
                         ("noslots", {"NGM_HIP_CS_NO_FIXED_SLOTS": "1"})):
            other = run(tag, personality, env)
            assert other == base, "%s %s: %d of %d lines differ" % (tag, personality, sum(a != b for a, b in zip(other, base)), len(base))


@needs_ref
def test_big_parity_60mbp(tmp_path):
    """tests/big_parity.py ('not collected by pytest'): 0 differing records in every mode."""
    env = dict(os.environ, BIG_SE="60000", BIG_PE="40000", BIG_ONLY="se,pe,pe-strata")
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "big_parity.py")], capture_output=True, text=True, env=env, timeout=3400)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if " differ;" in l]
    print("\n".join(lines))
    assert len(lines) == 3
    for l in lines:
        assert ", 0 differ;" in l, l
