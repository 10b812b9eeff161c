"""The GPU's BGZF blocks (nextgenmap_amd/csrc/bgzf_device.h) through the C-ABI: every block must be a BGZF member that zlib inflates
back to the input (bamtools' reader, samtools and python's gzip all read it that way), with the BC extra field's size right.
The oracle is zlib itself -- the library the reference writes its --bam output with (lib/bamtools-2.3.0 BgzfStream_p.cpp)."""
import gzip
import os
import random
import struct
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _members(z):
    """walk the BGZF members by their BSIZE fields: (offset, size, isize)"""
    out, at = [], 0
    while at < len(z):
        assert z[at:at + 4] == b"\x1f\x8b\x08\x04", at
        assert z[at + 10:at + 16] == b"\x06\x00BC\x02\x00", at
        size = struct.unpack_from("<H", z, at + 16)[0] + 1
        crc, isize = struct.unpack_from("<II", z, at + size - 8)
        raw = zlib.decompressobj(-15).decompress(z[at + 18:at + size - 8])
        assert len(raw) == isize and zlib.crc32(raw) == crc, at
        out.append((at, size, isize))
        at += size
    assert at == len(z)
    return out


def _bam_like(n, seed):
    rng = np.random.default_rng(seed)
    recs = []
    for i in range(n):
        name = b"read_%09d" % i + b"\0"
        seq = bytes(rng.choice(np.array([0x11, 0x12, 0x14, 0x18, 0x21, 0x22, 0x24, 0x28, 0x41, 0x42, 0x44, 0x48, 0x81, 0x82, 0x84, 0x88], np.uint8), 75))
        qual = bytes(rng.choice(np.array([37, 37, 37, 37, 25, 11, 2], np.uint8), 150))
        tags = b"NMi" + struct.pack("<i", int(rng.integers(0, 4))) + b"ASi" + struct.pack("<i", int(rng.integers(1200, 1500))) + b"XIf" + struct.pack("<f", 0.99) + b"MDZ150\0"
        body = struct.pack("<iiIIiiii", 3, int(rng.integers(0, 1 << 27)), (4681 << 16) | (60 << 8) | len(name), (99 << 16) | 1, 150, 3, int(rng.integers(0, 1 << 27)), 350) \
            + name + struct.pack("<I", 150 << 4) + seq + qual + tags
        recs.append(struct.pack("<I", len(body)) + body)
    return b"".join(recs)


def _cases():
    rnd = random.Random(3)
    c = {"one_byte": b"x", "three": b"abc", "zeros": bytes(200000), "random": os.urandom(150000), "run_258": b"q" * 258 + b"r" * 259 + b"s" * 600,
         "block_minus_1": os.urandom(100) * 700, "text": (b"the quick brown fox jumps over the lazy dog\n" * 5000)[:0xFF00 * 3 + 17],
         "short_periods": b"ab" * 40000 + b"abc" * 30000 + b"abcdefg" * 9000, "bam_like": _bam_like(4000, 1)}
    c["exact_block"] = c["bam_like"][:0xFF00]
    c["block_plus_1"] = c["bam_like"][:0xFF00 + 1]
    c["block_minus_1"] = c["bam_like"][:0xFF00 - 1]
    c["skewed"] = bytes(rnd.choices(range(256), weights=[2 ** (-(i % 50) / 2.2) for i in range(256)], k=300000))   # codes of up to 15 bits
    c["fib"] = b"".join(bytes([i]) * f for i, f in enumerate([1, 1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233, 377, 610, 987, 1597, 2584, 4181, 6765, 10946, 17711]))  # depth > 15 before the repair
    far = bytearray(os.urandom(20000))
    for _ in range(4000):
        a = rnd.randrange(0, len(far) - 300)
        far += far[a:a + rnd.randrange(3, 300)]
        far += os.urandom(rnd.randrange(0, 4))
    c["all_lengths_and_distances"] = bytes(far)
    return c


def test_blocks_inflate_to_the_input():
    from nextgenmap_amd.pipeline import Bgzf
    z = Bgzf(0)
    report = []
    for name, data in _cases().items():
        out = z.compress(data)
        assert gzip.decompress(out) == data, name
        mem = _members(out)
        assert [m[2] for m in mem] == [min(0xFF00, len(data) - i) for i in range(0, len(data), 0xFF00)], name
        assert z.compress(data) == out, name + " (not deterministic)"
        ref = sum(len(zlib.compress(data[i:i + 0xFF00], 6)) + 20 for i in range(0, len(data), 0xFF00))
        report.append("%-28s %8d -> %8d bytes (zlib level 6 per block: %8d; %.3f)" % (name, len(data), len(out), ref, len(out) / max(1, ref)))
    print("\n" + "\n".join(report))
    z.close()


def test_throughput_and_ratio_on_bam_records():
    import time
    from nextgenmap_amd.pipeline import Bgzf
    z = Bgzf(0)
    data = _bam_like(120000, 2)
    z.compress(data[:1 << 20])
    t = time.perf_counter()
    out = z.compress(data)
    dt = time.perf_counter() - t
    assert gzip.decompress(out) == data
    t6 = time.perf_counter()
    ref = sum(len(zlib.compress(data[i:i + 0xFF00], 6)) + 20 for i in range(0, len(data), 0xFF00))
    dt6 = time.perf_counter() - t6
    print("\n%d bytes of BAM-like records -> %d (zlib level 6: %d, ratio %.3f); kernel %.2f ms = %.1f GB/s, call %.1f ms; zlib level 6 on one core %.0f ms"
          % (len(data), len(out), ref, len(out) / ref, z.last_kernel_ms(), len(data) / z.last_kernel_ms() / 1e6, dt * 1e3, dt6 * 1e3))
    assert len(out) < 1.25 * ref
    z.close()
