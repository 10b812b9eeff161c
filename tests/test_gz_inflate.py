"""nextgenmap_amd/csrc/gz_inflate.h (the .gz reader of ngm-hip's input side) against zlib: every block type, codes of
every length, multi-member files, damaged files.  The reference reads .gz through zlib's gzread (src/parser/ReadProvider.cpp:240-262):
what gzread returns is the oracle here (python's gzip module = the same zlib)."""
import gzip
import os
import random
import subprocess
import zlib

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    d = tmp_path_factory.mktemp("gz")
    out = str(d / "gz_inflate_check")
    subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(ROOT, "tests", "cpp", "gz_inflate_check.cpp"), "-o", out, "-lz", "-lpthread"])
    return out, d


def _fastq(n, seed):
    rnd = random.Random(seed)
    out = []
    for i in range(n):
        s = "".join(rnd.choice("ACGT") for _ in range(150))
        q = "".join(rnd.choice("FFFFFFFFF:,#") for _ in range(150))
        out.append("@read_%09d/1\n%s\n+\n%s\n" % (i, s, q))
    return "".join(out).encode()


def _raw(data, **kw):
    c = zlib.compressobj(wbits=31, **kw)
    return c.compress(data) + c.flush()


def _cases():
    rnd = random.Random(11)
    fq = _fastq(12000, 5)
    cases = {}
    for lv in (1, 4, 6, 9):
        cases["fastq_level%d" % lv] = gzip.compress(fq, lv)
    cases["random_bytes_stored_blocks"] = gzip.compress(os.urandom(700000), 6)
    cases["zeros_distance_one"] = gzip.compress(bytes(3000000), 6)
    cases["tiny_fixed_huffman"] = gzip.compress(b"hello hello hello world\n", 6)
    cases["one_byte"] = gzip.compress(b"x", 6)
    cases["short_distances"] = gzip.compress(b"abc" * 300000 + b"xy" * 5000 + b"q" * 70000 + b"abcde" * 9000 + b"1234567" * 9000, 9)
    cases["fixed_huffman_blocks"] = _raw(fq[:900000], strategy=zlib.Z_FIXED)
    cases["huffman_only"] = _raw(fq[:900000], strategy=zlib.Z_HUFFMAN_ONLY)
    cases["run_length"] = _raw(fq[:900000], strategy=zlib.Z_RLE)
    cases["level0"] = _raw(fq[:1500000], level=0)
    cases["members"] = gzip.compress(fq[:600000], 6) + gzip.compress(b"", 6) + gzip.compress(fq[600000:1500000], 1) + gzip.compress(b"tail\n", 9)
    cases["many_small_members"] = b"".join(gzip.compress(fq[i:i + 60000], 6) for i in range(0, 1800000, 60000))
    c = zlib.compressobj(6, wbits=31)
    parts = []
    for i in range(0, 1500000, 100000):
        parts.append(c.compress(fq[i:i + 100000]))
        parts.append(c.flush(zlib.Z_FULL_FLUSH if i % 200000 else zlib.Z_SYNC_FLUSH))
    parts.append(c.flush())
    cases["flush_points"] = b"".join(parts)
    # codes of up to 15 bits in both trees: a geometric byte distribution, and matches at every distance class
    sk = bytes(rnd.choices(range(256), weights=[2 ** (-(i % 40) / 2.5) for i in range(256)], k=1500000))
    cases["long_codes"] = gzip.compress(sk, 6)
    far = bytearray(os.urandom(40000))
    for _ in range(3000):
        a = rnd.randrange(0, len(far) - 300)
        far += far[a:a + rnd.randrange(3, 259)]
        far += os.urandom(rnd.randrange(0, 6))
    cases["all_distances"] = gzip.compress(bytes(far), 9)
    c = zlib.compressobj(9, zlib.DEFLATED, 31, 9)
    cases["header_fields"] = b"\x1f\x8b\x08\x1c" + bytes(6) + b"\x04\x00abcd" + b"name.fq\0" + b"a comment\0" + (c.compress(fq[:300000]) + c.flush())[10:]
    return cases


def test_inflates_what_zlib_inflates(exe):
    prog, d = exe
    for name, z in _cases().items():
        p = str(d / (name + ".gz"))
        with open(p, "wb") as f:
            f.write(z)
        want = b"".join(gzip.decompress(z) for _ in [0])
        r = subprocess.run([prog, p, p + ".out"])
        assert r.returncode == 0, name
        with open(p + ".out", "rb") as f:
            assert f.read() == want, name


def test_trailing_garbage_is_ignored_like_gzread(exe):
    prog, d = exe
    text = _fastq(500, 2)
    p = str(d / "garbage.gz")
    with open(p, "wb") as f:
        f.write(gzip.compress(text, 6) + bytes(100))
    assert subprocess.run([prog, p, p + ".out"]).returncode == 0
    assert open(p + ".out", "rb").read() == text


def test_damaged_files_are_refused(exe):
    """a flipped bit in the stream, a wrong CRC, a wrong length, a truncated file: exit 3 -- ngm-hip then hands the file to zlib's reader,
    which reports the error the reference's user would see"""
    prog, d = exe
    text = _fastq(3000, 3)
    z = bytearray(gzip.compress(text, 6))
    rnd = random.Random(1)
    damaged = {"crc": bytes(z[:-8]) + bytes([z[-8] ^ 1]) + bytes(z[-7:]), "length": bytes(z[:-1]) + bytes([z[-1] ^ 1]),
               "truncated": bytes(z[:len(z) // 2]), "truncated_trailer": bytes(z[:-5])}
    for i in range(6):
        b = bytearray(z)
        b[rnd.randrange(40, len(z) - 40)] ^= 1 << rnd.randrange(8)
        damaged["bit_%d" % i] = bytes(b)
    for name, data in damaged.items():
        p = str(d / ("bad_" + name + ".gz"))
        with open(p, "wb") as f:
            f.write(data)
        assert subprocess.run([prog, p, p + ".out"]).returncode == 3, name


def _incomplete_literal_code_member():
    """a dynamic block whose literal/length alphabet has two codes of two bits (byte 0x00 and end-of-block): half the code space unused.
    zlib's inflate_table refuses it ("invalid literal/lengths set")."""
    bits = []

    def put(v, n):            # header fields: least significant bit first
        bits.extend((v >> i) & 1 for i in range(n))

    def code(v, n):           # Huffman codes: most significant bit first
        bits.extend((v >> (n - 1 - i)) & 1 for i in range(n))

    put(1, 1); put(2, 2)      # last block, dynamic codes
    put(0, 5); put(0, 5); put(12, 4)   # 257 literal/length codes, 1 distance code, 16 code-length code lengths
    order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2]
    pre = {18: 1, 0: 2, 2: 2}  # complete: 18 -> 0, 0 -> 10, 2 -> 11
    for s in order:
        put(pre.get(s, 0), 3)
    code(3, 2)                          # literal 0: length 2
    code(0, 1); put(138 - 11, 7)        # 138 zeros
    code(0, 1); put(117 - 11, 7)        # 117 zeros (literals 1..255)
    code(3, 2)                          # end of block: length 2
    code(2, 2)                          # the one distance code: unused
    code(0, 2); code(0, 2); code(1, 2)  # two bytes 0x00, end of block
    while len(bits) % 8:
        bits.append(0)
    body = bytes(sum(b << i for i, b in enumerate(bits[k:k + 8])) for k in range(0, len(bits), 8))
    text = b"\0\0"
    return b"\x1f\x8b\x08\x00" + bytes(6) + body + zlib.crc32(text).to_bytes(4, "little") + len(text).to_bytes(4, "little")


def test_incomplete_literal_code_is_refused_as_zlib_refuses_it(exe):
    prog, d = exe
    z = _incomplete_literal_code_member()
    with pytest.raises(zlib.error, match="invalid literal/lengths set"):
        zlib.decompress(z, 31)
    p = str(d / "incomplete_code.gz")
    with open(p, "wb") as f:
        f.write(z + bytes(40))   # (the checker wants a file of some size)
    assert subprocess.run([prog, p, p + ".out"]).returncode == 3


def test_corrupted_inputs_are_refused_or_decoded_like_zlib(tmp_path):
    """300 damaged files (flipped bits, truncation, overwritten and deleted spans) through a build with AddressSanitizer and
    UndefinedBehaviorSanitizer: the decoder's unchecked hot loop must stay inside its buffers whatever the stream says -- exit 3 (refused)
    or exit 0 with exactly zlib's text, never a crash."""
    exe = str(tmp_path / "gz_inflate_check_asan")
    c = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", os.path.join(ROOT, "tests", "cpp", "gz_inflate_check.cpp"), "-o", exe, "-lz", "-lpthread"],
                       capture_output=True, text=True)
    if c.returncode != 0:
        pytest.skip("no sanitizer runtime for g++ here: " + c.stderr[-200:])
    rnd = random.Random(7)
    base = [gzip.compress(_fastq(1500, 1), 1), gzip.compress(_fastq(1500, 2), 9), gzip.compress(os.urandom(30000), 6), gzip.compress(bytes(200000), 6)]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    seen = {0: 0, 3: 0}
    for it in range(300):
        z = bytearray(rnd.choice(base))
        mode = it % 4
        if mode == 0:
            for _ in range(rnd.randrange(1, 4)):
                z[rnd.randrange(len(z))] ^= 1 << rnd.randrange(8)
        elif mode == 1:
            z = z[:rnd.randrange(1, len(z))]
        elif mode == 2:
            a = rnd.randrange(len(z))
            z[a:a + rnd.randrange(1, 50)] = os.urandom(rnd.randrange(1, 50))
        else:
            a = rnd.randrange(10, len(z))
            z = z[:a] + z[a + rnd.randrange(1, 30):]
        p = str(tmp_path / "damaged.gz")
        with open(p, "wb") as f:
            f.write(bytes(z))
        r = subprocess.run([exe, p, p + ".out"], capture_output=True, env=env)
        assert r.returncode in (0, 3), (it, r.returncode, r.stderr[-800:])
        seen[r.returncode] += 1
        if r.returncode == 0:
            assert open(p + ".out", "rb").read() == gzip.decompress(bytes(z)), it
    assert seen[3] > 200
