"""Build recipe for the HIP library (gfx950 only).  `python -m nextgenmap_amd.build`"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
LIB = os.path.join(HERE, "libngm_hip.so")
SOURCES = ["ngm_hip.cpp", "jit.cpp", "ialignment_adapter.cpp", "refindex.cpp", "mapper.cpp", "mapper_search.cpp", "bgzf.cpp", "stats_reduce.cpp"]
JIT_HEADERS = ["sw_device.h", "align_device.h", "affine_device.h"]  # DP kernel templates, also compiled at run time (hiprtc)
OBJ_DIR = os.path.join(HERE, "build")


def _inputs():
    """every file the library or the CLI is built from (a hand-kept header list went stale)"""
    import glob
    inc = os.path.join(HERE, "..", "include")
    return sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.cpp")) + glob.glob(os.path.join(inc, "*.h")))


def _stale(target=None, inputs=None):
    target = target or LIB
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    for p in (inputs or _inputs()):
        if not p.endswith("jit_sources.inc") and os.path.getmtime(p) > t:
            return True
    return False


def write_jit_sources():
    """csrc/jit_sources.inc: the DP kernel headers as one string for hiprtc (the reference JIT-compiles its kernels with
    -D corridor_length; here corridors without an ahead-of-time build are compiled on first use)."""
    parts = []
    for h in JIT_HEADERS:
        for line in open(os.path.join(CSRC, h)).read().splitlines():
            if line.startswith("#include") or line.startswith("#pragma once"):
                continue
            parts.append(line)
    text = "\n".join(parts) + "\n"
    out = os.path.join(CSRC, "jit_sources.inc")
    chunks = [text[i:i + 8000] for i in range(0, len(text), 8000)]  # keep every literal well below compiler limits
    body = "static const char *const kJitSourceChunks[] = {\n" + ",\n".join('R"NGMJIT(' + c + ')NGMJIT"' for c in chunks) + "\n};\n"
    if not os.path.exists(out) or open(out).read() != body:
        open(out, "w").write(body)
    return out


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU), one translation unit per process. Returns the library path."""
    from concurrent.futures import ThreadPoolExecutor
    write_jit_sources()
    if not force and not _stale():
        if _stale(CLI, [os.path.join(CSRC, "ngm_cli.cpp"), LIB] + [p for p in _inputs() if p.endswith(".h")]):
            build_cli(verbose)
        return LIB
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = [p for p in _inputs() if p.endswith(".h")] + [os.path.join(CSRC, "jit_sources.inc")]

    def compile_one(src):
        obj = os.path.join(OBJ_DIR, src.replace(".cpp", ".o"))
        path = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and all(os.path.getmtime(obj) >= os.path.getmtime(d) for d in [path] + headers):
            return obj
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-lz", "-lhiprtc", "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    build_cli(verbose)
    return LIB


CLI = os.path.join(HERE, "ngm-hip")


def build_cli(verbose=False):
    """The NextGenMap-compatible command line: plain g++ host program over the C ABI of the library."""
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-g1", "-std=c++17", os.path.join(CSRC, "ngm_cli.cpp"), LIB, "-lz", "-ldl",
           "-Wl,-rpath," + HERE, "-Wl,-rpath,/opt/rocm/lib", "-o", CLI]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return CLI


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
