"""Build recipe for the HIP library (gfx950 only).  `python -m nextgenmap_amd.build`"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
LIB = os.path.join(HERE, "libngm_hip.so")
SOURCES = ["ngm_hip.cpp", "ialignment_adapter.cpp", "refindex.cpp", "mapper.cpp"]
HEADERS = ["ngm_cli.cpp", "engine_internal.h", "sw_device.h", "align_device.h", "cigar_md.h", "refindex.h", "cs_device.h", "gather_device.h", os.path.join("..", "..", "include", "ngm_pipeline.h"), os.path.join("..", "..", "include", "ngm_hip.h"),
           os.path.join("..", "..", "include", "ngm_ialignment.h")]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU). Returns the library path."""
    if not force and not _stale():
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip"] + srcs + ["-lz", "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    build_cli(verbose)
    return LIB


CLI = os.path.join(HERE, "ngm-hip")


def build_cli(verbose=False):
    """The NextGenMap-compatible command line: plain g++ host program over the C ABI of the library."""
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", os.path.join(CSRC, "ngm_cli.cpp"), LIB, "-lz",
           "-Wl,-rpath," + HERE, "-Wl,-rpath,/opt/rocm/lib", "-o", CLI]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return CLI


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
