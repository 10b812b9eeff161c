"""ctypes bindings for the TEST-ONLY checkers under oracle/ (the C restatement and, on a GPU box,
the driver for the reference's own kernels).  Never imported by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


class Scoring(C.Structure):
    _fields_ = [("match", C.c_int), ("mismatch", C.c_int), ("gap_read", C.c_int), ("gap_ref", C.c_int),
                ("alt", C.c_int), ("match_alt", C.c_int), ("mismatch_alt", C.c_int), ("dir", C.c_int)]


class Trace(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("valid", "best_read_index", "best_ref_index", "ref_position",
                                       "qstart", "qend", "alignment_offset", "best_score")]


class AlignRes(C.Structure):
    _fields_ = [("ok", C.c_int), ("position_offset", C.c_int), ("qstart", C.c_int), ("qend", C.c_int),
                ("nm", C.c_int), ("identity", C.c_float), ("score_token", C.c_float)]


ALIGN_DTYPE = np.dtype([("ok", "i4"), ("position_offset", "i4"), ("qstart", "i4"), ("qend", "i4"),
                        ("nm", "i4"), ("identity", "f4"), ("score_token", "f4")])

DEFAULT_SCORING = dict(match=10, mismatch=-15, gap_read=-20, gap_ref=-20)
# `ngm --bs-mapping` (src/config/Config.cpp:461-467; matchALT = MATCH_BONUS_TT, mismatchALT = MATCH_BONUS_TC) and the
# scores of a `--slam-seq 2` run (Config.cpp:433-447; mismatchALT = -MATCH_BONUS_TC, lib/mason/opencl/SWOcl.cpp:233-238)
BS_SCORING = dict(match=4, mismatch=-2, gap_read=-10, gap_ref=-10, alt=1, match_alt=4, mismatch_alt=4)
SLAM_SCORING = dict(match=10, mismatch=-15, gap_read=-20, gap_ref=-20, alt=2, match_alt=10, mismatch_alt=-2)


def _build(target):
    subprocess.check_call(["make", "-C", ORACLE_DIR, target], stdout=subprocess.DEVNULL)


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        path = os.path.join(ORACLE_DIR, "libngm_oracle.so")
        if not os.path.exists(path):
            _build("libngm_oracle.so")
        lib = C.CDLL(path)
        lib.ngm_oracle_batch_score.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_long,
                                               C.c_int, C.c_int, C.POINTER(Scoring), C.c_int, C.c_void_p, C.c_int]
        lib.ngm_oracle_batch_align.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_long,
                                               C.c_int, C.c_int, C.POINTER(Scoring), C.c_int, C.c_int, C.c_int,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int]
        lib.ngm_oracle_batch_score_alt.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_long,
                                                   C.c_int, C.c_int, C.POINTER(Scoring), C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        lib.ngm_oracle_batch_align_alt.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_long,
                                                   C.c_int, C.c_int, C.POINTER(Scoring), C.c_int, C.c_void_p, C.c_int, C.c_int,
                                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int]
        lib.ngm_oracle_align_trace.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                               C.POINTER(Scoring), C.c_int, C.POINTER(Trace), C.c_void_p]
        _oracle = lib
    return _oracle


def _sc(scoring):
    s = dict(DEFAULT_SCORING)
    if scoring:
        s.update(scoring)
    return Scoring(**s)


def _dirs(dirs, n):
    if dirs is None:
        return None, None
    d = np.ascontiguousarray(dirs, dtype=np.uint8)
    assert d.shape == (n,)
    return d, d.ctypes.data



def golden_case(g):
    """(oracle scoring dict or None, direction bytes or None, Engine keyword arguments) of a golden file written by
    oracle/make_goldens.py: the -D__ALT_SCORING__ cases carry `alt`, `scoring` and `dirs`."""
    if "alt" not in g.files:
        return None, None, {}
    v = [int(x) for x in g["scoring"]]
    sc = dict(zip(("match", "mismatch", "gap_read", "gap_ref", "alt", "match_alt", "mismatch_alt"), v))
    kw = dict(match=sc["match"], mismatch=-sc["mismatch"], gap_read=-sc["gap_read"], gap_ref=-sc["gap_ref"], alt_scoring=sc["alt"],
              match_bonus_tt=sc["match_alt"], match_bonus_tc=abs(sc["mismatch_alt"]))
    return sc, g["dirs"], kw


def oracle_score(mode, ref, qry, c, scoring=None, variant=0, nthreads=1, dirs=None):
    """ref [n, q+c] uint8, qry [n, q] uint8 -> float32[n]; dirs: the per-pair direction bytes of the ALT scoring modes"""
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    qry = np.ascontiguousarray(qry, dtype=np.uint8)
    n, q = qry.shape
    assert ref.shape == (n, q + c)
    out = np.empty(n, dtype=np.float32)
    sc = _sc(scoring)
    d, dp = _dirs(dirs, n)
    oracle().ngm_oracle_batch_score_alt(mode, n, ref.ctypes.data, ref.strides[0], qry.ctypes.data, qry.strides[0],
                                        q, c, C.byref(sc), variant, dp, out.ctypes.data, nthreads)
    return out


def oracle_align(mode, ref, qry, c, scoring=None, variant=0, hard_clip=0, silent_clip=0, nthreads=1, dirs=None):
    """-> (structured array ALIGN_DTYPE [n], cigars list[bytes], mds list[bytes])"""
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    qry = np.ascontiguousarray(qry, dtype=np.uint8)
    n, q = qry.shape
    stride = 4 * max(1, q)
    res = np.zeros(n, dtype=ALIGN_DTYPE)
    cig = np.zeros((n, stride), dtype=np.uint8)
    md = np.zeros((n, stride), dtype=np.uint8)
    sc = _sc(scoring)
    d, dp = _dirs(dirs, n)
    oracle().ngm_oracle_batch_align_alt(mode, n, ref.ctypes.data, ref.strides[0], qry.ctypes.data, qry.strides[0],
                                        q, c, C.byref(sc), variant, dp, hard_clip, silent_clip, res.ctypes.data,
                                        cig.ctypes.data, md.ctypes.data, stride, nthreads)
    cigs = [bytes(r).split(b"\0", 1)[0] for r in cig]
    mds = [bytes(r).split(b"\0", 1)[0] for r in md]
    return res, cigs, mds


def oracle_trace(mode, ref, qry, c, scoring=None, variant=0, dirs=None):
    """Raw kernel-level outputs: (results4 int16 [n,4] as the reference leaves them, rle int16 [n, 2*AL],
    valid bool[n], best_score int32[n])."""
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    qry = np.ascontiguousarray(qry, dtype=np.uint8)
    n, q = qry.shape
    al = 2 * q + c + 1
    res = np.zeros((n, 4), dtype=np.int16)
    rle = np.zeros((n, 2 * al), dtype=np.int16)
    valid = np.zeros(n, dtype=bool)
    best = np.zeros(n, dtype=np.int32)
    sc = _sc(scoring)
    tr = Trace()
    lib = oracle()
    for i in range(n):
        sc.dir = int(dirs[i] != 0) if dirs is not None else 0
        lib.ngm_oracle_align_trace(mode, ref[i].ctypes.data, qry[i].ctypes.data, q, c, C.byref(sc), variant,
                                   C.byref(tr), rle[i].ctypes.data)
        valid[i] = bool(tr.valid)
        best[i] = tr.best_score
        if tr.valid:
            res[i] = (tr.ref_position, tr.qstart, tr.qend, tr.alignment_offset)
        else:
            res[i] = (tr.best_read_index, tr.best_ref_index, tr.qend, 0)
    return res, rle, valid, best


# ---------------------------------------------------------------------------------------------
# affine personality: restatement + the reference's own EndToEndAffine behind oracle/_ref/ngm/ngm_affine_ref
# ---------------------------------------------------------------------------------------------
class AffScoring(C.Structure):
    _fields_ = [("match", C.c_int), ("mismatch", C.c_int), ("gap_open", C.c_int), ("gap_extend", C.c_int)]


DEFAULT_AFFINE = dict(match=10, mismatch=-15, gap_open=-33, gap_extend=-3)
AFFINE_REF = os.path.join(ORACLE_DIR, "_ref", "ngm", "ngm_affine_ref")


def oracle_affine(mode, ref, qry, c, scoring=None, nthreads=1):
    """-> (scores float32[n], ALIGN_DTYPE[n], cigars list[bytes])"""
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    qry = np.ascontiguousarray(qry, dtype=np.uint8)
    n, q = qry.shape
    s = dict(DEFAULT_AFFINE)
    if scoring:
        s.update(scoring)
    sc = AffScoring(**s)
    lib = oracle()
    lib.ngm_oracle_affine_batch.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_int,
                                            C.POINTER(AffScoring), C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int]
    stride = 4 * max(1, q)
    scores = np.zeros(n, np.float32)
    res = np.zeros(n, dtype=ALIGN_DTYPE)
    cig = np.zeros((n, stride), np.uint8)
    # rows must be NUL terminated inside q+c / q bytes: pad one column
    refp = np.zeros((n, ref.shape[1] + 1), np.uint8); refp[:, :-1] = ref
    qryp = np.zeros((n, q + 1), np.uint8); qryp[:, :-1] = qry
    lib.ngm_oracle_affine_batch(mode, n, refp.ctypes.data, refp.strides[0], qryp.ctypes.data, qryp.strides[0], q, c, C.byref(sc),
                                scores.ctypes.data, res.ctypes.data, cig.ctypes.data, stride, nthreads)
    return scores, res, [bytes(r).split(b"\0", 1)[0] for r in cig]


def reference_affine(mode, ref, qry, c, scoring=None, workdir="/tmp"):
    """Run the REFERENCE's EndToEndAffine (SeqAn) on the pairs. -> list of None (empty pair) or
    (score, cigar, position_offset, qstart, qend, nm, identity)."""
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    qry = np.ascontiguousarray(qry, dtype=np.uint8)
    n, q = qry.shape
    s = dict(DEFAULT_AFFINE)
    if scoring:
        s.update(scoring)
    import tempfile
    with tempfile.TemporaryDirectory(dir=workdir) as d:
        inp, outp = os.path.join(d, "in.bin"), os.path.join(d, "out.txt")
        with open(inp, "wb") as f:
            np.array([n, q, c], np.int32).tofile(f)
            ref.tofile(f)
            qry.tofile(f)
        r = subprocess.run([AFFINE_REF, inp, outp, str(mode), str(s["match"]), str(-s["mismatch"]), str(-s["gap_open"]),
                            str(-s["gap_extend"])], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("affine reference failed: " + r.stderr[-500:])
        out = []
        for line in open(outp):
            f = line.rstrip("\n").split("\t")
            if f[1] == "EMPTY":
                out.append(None)
            else:
                out.append((int(f[1]), f[2].encode(), int(f[3]), int(f[4]), int(f[5]), int(f[6]), np.float32(float(f[7]))))
    return out


# ---------------------------------------------------------------------------------------------
# reference kernels on the GPU (oracle/_ref/*.co via oracle/libngm_ref_runner.so)
# ---------------------------------------------------------------------------------------------
def ref_co_path(variant, q, c, scoring=None):
    s = dict(DEFAULT_SCORING)
    if scoring:
        s.update(scoring)
    name = "ngm_ocl_%s_q%d_c%d_m%d_x%d_gr%d_gf%d" % (("gpu", "cpu", "gpu1")[variant], q, c, s["match"],
                                                      -s["mismatch"], -s["gap_read"], -s["gap_ref"])
    if s.get("alt") == 1:     # oracle/build_ref.sh ... bs <tt> <tc>
        name += "_bs_tt%d_tc%d" % (s["match_alt"], s["mismatch_alt"])
    elif s.get("alt") == 2:   # ... slam <tt> <tc>  (mismatchALT = -tc)
        name += "_slam_tt%d_tc%d" % (s["match_alt"], -s["mismatch_alt"])
    return os.path.join(ORACLE_DIR, "_ref", name + ".co")


_runner = None


def ref_runner():
    global _runner
    if _runner is None:
        path = os.path.join(ORACLE_DIR, "libngm_ref_runner.so")
        lib = C.CDLL(path)
        lib.ngm_ref_run_score.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_int, C.c_void_p, C.POINTER(C.c_float)]
        lib.ngm_ref_run_align.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
        lib.ngm_ref_run_score_alt.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
        lib.ngm_ref_run_align_alt.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
        _runner = lib
    return _runner


def ref_score(variant, mode, ref, qry, c, scoring=None, dirs=None):
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    qry = np.ascontiguousarray(qry, dtype=np.uint8)
    n, q = qry.shape
    out = np.empty(n, dtype=np.float32)
    ms = C.c_float(0)
    co = ref_co_path(variant, q, c, scoring)
    if scoring and scoring.get("alt"):   # a -D__ALT_SCORING__ build: its kernels take the direction bytes
        d = np.zeros(n, np.uint8) if dirs is None else np.ascontiguousarray(dirs, dtype=np.uint8)
        r = ref_runner().ngm_ref_run_score_alt(co.encode(), variant, mode, n, ref.ctypes.data, qry.ctypes.data, q, c, d.ctypes.data,
                                               out.ctypes.data, C.byref(ms))
    else:
        r = ref_runner().ngm_ref_run_score(co.encode(), variant, mode, n, ref.ctypes.data, qry.ctypes.data, q, c,
                                           out.ctypes.data, C.byref(ms))
    if r != n:
        raise RuntimeError("reference kernel run failed (%s)" % co)
    return out, ms.value


def ref_align(variant, mode, ref, qry, c, scoring=None, dirs=None):
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    qry = np.ascontiguousarray(qry, dtype=np.uint8)
    n, q = qry.shape
    al = 2 * q + c + 1
    res = np.zeros((n, 4), dtype=np.int16)
    rle = np.zeros((n, 2 * al), dtype=np.int16)
    ms = C.c_float(0)
    co = ref_co_path(variant, q, c, scoring)
    if scoring and scoring.get("alt"):
        d = np.zeros(n, np.uint8) if dirs is None else np.ascontiguousarray(dirs, dtype=np.uint8)
        r = ref_runner().ngm_ref_run_align_alt(co.encode(), variant, mode, n, ref.ctypes.data, qry.ctypes.data, q, c, d.ctypes.data,
                                               res.ctypes.data, rle.ctypes.data, C.byref(ms))
    else:
        r = ref_runner().ngm_ref_run_align(co.encode(), variant, mode, n, ref.ctypes.data, qry.ctypes.data, q, c,
                                           res.ctypes.data, rle.ctypes.data, C.byref(ms))
    if r != n:
        raise RuntimeError("reference kernel run failed (%s)" % co)
    return res, rle, ms.value
