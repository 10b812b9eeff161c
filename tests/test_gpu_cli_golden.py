"""-m gpu: `ngm-hip` against committed golden SAM files captured from the REAL reference program (`ngm --affine`,
oracle/make_cli_goldens.py): no reference binary needed at test time.  ngm-hip reads the gzipped FASTA / FASTQ directly."""
import gzip
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "nextgenmap_amd", "ngm-hip")
GOLD = os.path.join(ROOT, "tests", "golden", "cli")


def _records(lines):
    out = {}
    for line in lines:
        if line.startswith("@"):
            continue
        f = line.rstrip("\n").split("\t")
        out.setdefault((f[0], int(f[1]) & 0xC0), []).append(tuple(f[1:]))
    return {k: sorted(v) for k, v in out.items()}


@pytest.mark.parametrize("case,args", [("pe_odd", ["-p", "-q", "pe_odd.fq.gz", "-X", "420", "--no-unal", "-R", "0.6"]), ("mixed_fasta", ["-q", "mixed.fa.gz"]), ("se_local", ["-q", "se.fq.gz"]), ("se_endtoend", ["-q", "se.fq.gz", "-e"]),
                                       ("se_top3", ["-q", "se.fq.gz", "-n", "3"]), ("pe_local", ["-p", "-q", "pe.fq.gz"])])
def test_sam_equals_golden_reference_output(tmp_path, case, args):
    from nextgenmap_amd import build
    build.build()
    argv = [a if not a.endswith(".gz") else os.path.join(GOLD, a) for a in args]
    out = str(tmp_path / "out.sam")
    c = subprocess.run([CLI, "-r", os.path.join(GOLD, "ref.fa.gz"), "-o", out, "--affine", "--skip-save"] + argv, capture_output=True, text=True)
    assert c.returncode == 0, c.stderr[-2000:]
    want_lines = gzip.open(os.path.join(GOLD, case + ".sam.gz"), "rt").read().splitlines(True)
    got_lines = open(out).read().splitlines(True)
    assert [l for l in want_lines if l.startswith("@SQ")] == [l for l in got_lines if l.startswith("@SQ")]
    want, got = _records(want_lines), _records(got_lines)
    assert set(want) == set(got)
    diff = [(k, want[k], got[k]) for k in want if want[k] != got[k]]
    assert not diff, (len(diff), str(diff[:2])[:1500])
