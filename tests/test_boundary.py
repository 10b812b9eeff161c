"""The drop-in boundary: exported symbols, header/ABI agreement with the reference's interfaces, and
(on the GPU) a C++ host that drives the plugin exactly as NextGenMap's ScoreBuffer/AlignmentBuffer do."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O
from pairgen import make_pairs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")
CPP = os.path.join(ROOT, "tests", "cpp")
REFERENCE = os.environ.get("NGM_REFERENCE", "/root/reference")


def _lib():
    from nextgenmap_amd import build
    return build.build()


def test_library_exports_every_declared_symbol():
    """Every function include/ngm_hip.h and include/ngm_ialignment.h declare is exported (no compute)."""
    lib = _lib()
    syms = set(re.findall(r" T (\w+)", subprocess.check_output(["nm", "-D", "--defined-only", lib]).decode()))
    hdr = open(os.path.join(INC, "ngm_hip.h")).read()
    declared = set(re.findall(r"\b(ngm_hip_\w+)\s*\(", hdr))
    declared -= {"ngm_hip_ctx"}
    assert len(declared) >= 13
    assert declared <= syms, "missing: %s" % sorted(declared - syms)
    plugin = {"SetLog", "SetConfig", "Cookie", "IsAvailable", "CreateAlignment", "DeleteAlignment", "ExternalDeleteString"}
    assert plugin <= syms
    # ... and the pipeline ABI (reference index, candidate search, mapping): include/ngm_pipeline.h
    hdr2 = re.sub(r"/\*.*?\*/", "", open(os.path.join(INC, "ngm_pipeline.h")).read(), flags=re.S)
    declared2 = set(re.findall(r"\b(ngm_(?:ref|mapper|pair_state|host|pipeline)_\w+)\s*\(", hdr2))
    assert len(declared2) >= 25, sorted(declared2)
    assert declared2 <= syms, "missing: %s" % sorted(declared2 - syms)


def test_library_loads_and_fails_loudly_without_gpu():
    import nextgenmap_amd as N
    lib = N.load_library()
    if lib.ngm_hip_device_count() == 0:
        with pytest.raises(N.NgmHipError, match="no HIP device"):
            N.Engine(152, 27)


def test_bad_parameters_are_rejected():
    import ctypes as C
    import nextgenmap_amd as N
    from nextgenmap_amd.engine import Params
    lib = N.load_library()
    for p in (Params(99, 152, 27, 10, 15, 20, 20, 0, 0, 0, 0), Params(1, 152, 27, 0, 15, 20, 20, 0, 0, 0, 0),
              Params(1, 152, 1, 10, 15, 20, 20, 0, 0, 0, 0), Params(1, 152, 27, 200, 100, 20, 20, 0, 0, 0, 0)):
        assert not lib.ngm_hip_create(0, C.byref(p))
        assert lib.ngm_hip_last_error(None)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "include")), reason="reference tree not present")
def test_vtable_and_struct_abi_match_reference_headers(tmp_path):
    """A consumer compiled against the reference's IAlignment.h/IConfig.h/ILog.h calls objects implemented
    against include/ngm_ialignment.h: every virtual must land on the right slot."""
    exe = str(tmp_path / "abi_check")
    o1, o2 = str(tmp_path / "c.o"), str(tmp_path / "m.o")
    subprocess.check_call(["g++", "-std=gnu++11", "-w", "-c", os.path.join(CPP, "abi_consumer_ref.cpp"),
                           "-I", os.path.join(REFERENCE, "include"), "-o", o1])
    subprocess.check_call(["g++", "-std=gnu++11", "-w", "-c", os.path.join(CPP, "abi_impl_mine.cpp"), "-I", INC, "-o", o2])
    subprocess.check_call(["g++", o1, o2, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1])
def test_cpp_host_drives_plugin_like_ngm(tmp_path, mode):
    lib = _lib()
    exe = str(tmp_path / "ngm_host_driver")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(CPP, "ngm_host_driver.cpp"), "-I", INC, lib,
                           "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    q, c, n = 152, 27, 777
    ref, qry = make_pairs(n, q, c, seed=31 + mode, read_len=150)
    inp = str(tmp_path / "in.bin")
    with open(inp, "wb") as f:
        np.array([n, q, c], dtype=np.int32).tofile(f)
        ref.tofile(f)
        qry.tofile(f)
    outp = str(tmp_path / "out.txt")
    r = subprocess.run([exe, inp, outp, str(mode)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    want_s = O.oracle_score(mode, ref, qry, c)
    res, cig, md = O.oracle_align(mode, ref, qry, c)
    rows = [l.rstrip("\n").split("\t") for l in open(outp)]
    assert len(rows) == n
    for i, row in enumerate(rows):
        assert int(row[1]) == int(want_s[i])
        if not res["ok"][i]:
            assert int(row[9]) == -1 and row[2] == "!!!"
            continue
        assert row[2].encode() == cig[i] and row[3].encode() == md[i]
        assert (int(row[4]), int(row[5]), int(row[6]), int(row[7])) == (
            int(res["position_offset"][i]), int(res["qstart"][i]), int(res["qend"][i]), int(res["nm"][i]))
        assert np.float32(float(row[8])) == np.float32(res["identity"][i])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1])
def test_cpp_host_drives_plugin_in_the_affine_personality(tmp_path, mode):
    """Config "affine" = 1 makes the same plugin exports behave like EndToEndAffine (the class NextGenMap instantiates for
    `--affine`, src/NGM.cpp:397-404): scores, CIGARs, NM, identity as SeqAn's banded Gotoh, pBuffer2 untouched."""
    lib = _lib()
    exe = str(tmp_path / "ngm_host_driver")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(CPP, "ngm_host_driver.cpp"), "-I", INC, lib,
                           "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    q, c, n = 152, 27, 700
    ref, qry = make_pairs(n, q, c, seed=41 + mode, read_len=150)
    inp = str(tmp_path / "in.bin")
    with open(inp, "wb") as f:
        np.array([n, q, c], dtype=np.int32).tofile(f)
        ref.tofile(f)
        qry.tofile(f)
    outp = str(tmp_path / "out.txt")
    r = subprocess.run([exe, inp, outp, str(mode), "affine"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    sc, res, cig = O.oracle_affine(mode, ref, qry, c, nthreads=4)
    rows = [l.rstrip("\n").split("\t") for l in open(outp)]
    assert len(rows) == n
    for i, row in enumerate(rows):
        assert int(row[1]) == int(sc[i])
        assert row[2].encode() == cig[i] and row[3] == "!!!"
        assert (int(row[4]), int(row[5]), int(row[6]), int(row[7])) == (
            int(res["position_offset"][i]), int(res["qstart"][i]), int(res["qend"][i]), int(res["nm"][i]))
        wi, hi = np.float32(res["identity"][i]), np.float32(float(row[8]))
        assert wi == hi or (np.isnan(wi) and np.isnan(hi))


def test_runtime_kernel_compilation_for_other_corridors():
    """Band widths without an ahead-of-time build are compiled with hiprtc from the embedded kernel headers (the
    reference JIT-compiles its OpenCL kernels with -D corridor_length); compiling needs no GPU."""
    import ctypes as C
    from nextgenmap_amd import build
    lib = C.CDLL(build.build())
    lib.ngm_hip_jit_selftest.restype = C.c_long
    msg = C.create_string_buffer(4096)
    size = lib.ngm_hip_jit_selftest(23, msg, 4096)
    assert size > 10000, msg.value.decode()
    assert b"sw_score_kernelILi23E" in msg.value
