"""CPU-only tests of the header-only host pieces of the product (no GPU, no library): the BAM / BGZF writer behind `ngm-hip --bam`
(csrc/bam_writer.h; the GPU tier compares whole files with the reference program's, tests/test_gpu_bam.py) and the thread pool
every host stage runs on (csrc/thread_pool.h)."""
import gzip
import os
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("host_units") / "host_units_driver")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", os.path.join(ROOT, "tests", "cpp", "host_units_driver.cpp"), "-lz", "-o", exe])
    return exe


def reg2bin(beg, end):  # SAM specification, section 5.3
    end -= 1
    for shift, off in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return off + (beg >> shift)
    return 0


def test_bam_writer_produces_a_valid_file(driver, tmp_path):
    out = str(tmp_path / "t.bam")
    r = subprocess.run([driver, "bam", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = open(out, "rb").read()
    # BGZF: gzip members with the BC extra field and their own size; the last one is the 28-byte end-of-file marker
    assert raw[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    at, members = 0, 0
    while at < len(raw):
        assert raw[at:at + 4] == b"\x1f\x8b\x08\x04" and raw[at + 12:at + 16] == b"BC\x02\x00"
        bsize, = struct.unpack_from("<H", raw, at + 16)
        isize, = struct.unpack_from("<I", raw, at + bsize + 1 - 4)
        assert isize <= 0xFF00
        at += bsize + 1
        members += 1
    assert at == len(raw) and members >= 5  # header, 3 record chunks (several blocks each), EOF
    data = gzip.decompress(raw)
    assert data[:4] == b"BAM\1"
    l_text, = struct.unpack_from("<i", data, 4)
    assert data[8:8 + l_text].decode().startswith("@HD\tVN:1.0")
    p = 8 + l_text
    n_ref, = struct.unpack_from("<i", data, p)
    p += 4
    refs = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", data, p)
        refs.append((data[p + 4:p + 4 + l_name - 1].decode(), struct.unpack_from("<i", data, p + 4 + l_name)[0]))
        p += 8 + l_name
    assert refs == [("chr1", 1000000), ("chrTwo", 54321)]
    n = 0
    while p < len(data):
        block, = struct.unpack_from("<i", data, p)
        ref_id, pos, bin_mq_nl, flag_nc, l_seq, mate_ref, mate_pos, tlen = struct.unpack_from("<iiIIiiii", data, p + 4)
        l_name, n_cig = bin_mq_nl & 0xFF, flag_nc & 0xFFFF
        q = p + 36
        assert data[q:q + l_name] == b"read%05d\0" % n
        q += l_name
        ops = struct.unpack_from("<%dI" % n_cig, data, q)
        q += 4 * n_cig
        unmapped = n % 97 == 0
        want_len = 100 + n % 51
        assert l_seq == want_len
        if unmapped:
            assert (ref_id, pos, n_cig, flag_nc >> 16) == (-1, -1, 0, 4)
            ref_span = 0
        else:
            lead = n % 7 + 1
            assert [(o >> 4, "MIDNSHP=X"[o & 15]) for o in ops] == [(lead, "S"), (20, "M"), (2, "I"), (30, "M"), (1, "D"), (want_len - 53 - lead, "M")]
            assert (ref_id, pos, (bin_mq_nl >> 8) & 0xFF, flag_nc >> 16) == (n % 2, 1000 + 37 * n, 60, 16 if n % 2 else 0)
            ref_span = 20 + 30 + 1 + want_len - 53 - lead
        assert bin_mq_nl >> 16 == reg2bin(pos, pos + ref_span)  # (bamtools' CalculateMinimumBin, also for the unmapped -1 / -1)
        seq = data[q:q + (l_seq + 1) // 2]
        q += (l_seq + 1) // 2
        want = "".join("ACGTN"[(n + 3 * t) % 5] for t in range(l_seq))
        got = "".join("=ACMGRSVTWYHKDBN"[(seq[t // 2] >> (4 if t % 2 == 0 else 0)) & 15] for t in range(l_seq))
        assert got == want
        qual = data[q:q + l_seq]
        q += l_seq
        assert qual == (bytes([ord(":") - 33]) * l_seq if n % 5 == 0 else bytes((n + t) % 40 for t in range(l_seq)))
        tags = data[q:p + 4 + block]
        assert tags == b"ASi" + struct.pack("<i", 1234 - n) + b"NMi" + struct.pack("<i", n % 9) + b"XIf" + struct.pack("<f", 0.9876) + b"MDZ50A49\0"
        assert (mate_ref, mate_pos, tlen) == (-1, -1, 0)
        p += 4 + block
        n += 1
    assert n == 2100
    # CalculateMinimumBin: the smallest bin that holds [begin, end)
    assert r.stdout.split()[1:] == [str(reg2bin(0, 1)), str(reg2bin(16383, 16385)), str(reg2bin(1 << 20, (1 << 20) + 150)), str(reg2bin(-1, -1)), str(reg2bin(100000000, 100000200))]


def test_thread_pool_covers_every_index_once_also_with_concurrent_and_nested_callers(driver):
    r = subprocess.run([driver, "pool"], capture_output=True, text=True, timeout=300, env=dict(os.environ, NGM_HIP_HOST_THREADS="8"))
    assert r.returncode == 0, r.stdout + r.stderr


def test_cli_rejects_what_it_does_not_support_before_touching_a_gpu():
    """`ngm-hip` takes NextGenMap's option names (src/config/Options.h); options of modes that are not built (bisulfite, SLAM-seq,
    fast pairing, argos, vcf) are refused loudly instead of being ignored, and so are unknown ones."""
    from nextgenmap_amd import build
    build.build()
    for bad in (["--bs-mapping"], ["--slam-seq", "2"], ["--frobnicate"]):
        r = subprocess.run([build.CLI, "-r", "x.fa", "-q", "y.fq", "-o", "/dev/null"] + bad, capture_output=True, text=True)
        assert r.returncode != 0, bad
        assert "[ngm-hip] error" in r.stderr, (bad, r.stderr)
    r = subprocess.run([build.CLI, "-r", "/nonexistent/ref.fa", "-q", "y.fq", "-o", "/dev/null"], capture_output=True, text=True)
    assert r.returncode != 0 and "cannot open reference" in (r.stdout + r.stderr)
