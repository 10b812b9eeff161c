"""ctypes mirror of the IAlignment plugin surface over the C ABI (include/ngm_hip.h).

Method names follow NextGenMap's interface (include/IAlignment.h:49-69): GetScoreBatchSize,
GetAlignBatchSize, BatchScore, BatchAlign -- same argument meaning, same "returns number processed"
contract.  The *_device methods take HBM-resident flat batches (torch CUDA tensors or raw pointers).
"""
import ctypes as C
import os

import numpy as np

MODE_LOCAL = 0
MODE_END_TO_END = 1
VARIANT_OCL_GPU = 0
VARIANT_OCL_CPU = 1
PERSONALITY_LINEAR = 0   # the OpenCL plugin (default)
PERSONALITY_AFFINE = 1   # `ngm --affine`: EndToEndAffine over SeqAn's banded Gotoh alignment
ABI_VERSION = 3

_HERE = os.path.dirname(os.path.abspath(__file__))


class NgmHipError(RuntimeError):
    pass


class Params(C.Structure):
    _fields_ = [("abi_version", C.c_int), ("qry_max_len", C.c_int), ("corridor", C.c_int),
                ("match_bonus", C.c_int), ("mismatch_penalty", C.c_int), ("gap_read_penalty", C.c_int),
                ("gap_ref_penalty", C.c_int), ("variant", C.c_int), ("hard_clip", C.c_int),
                ("silent_clip", C.c_int), ("max_batch", C.c_int), ("personality", C.c_int),
                ("gap_extend_penalty", C.c_int), ("alt_scoring", C.c_int), ("match_bonus_tt", C.c_int), ("match_bonus_tc", C.c_int), ("alt_cigar", C.c_int)]


class AlignOut(C.Structure):
    _fields_ = [("cigar", C.c_void_p), ("md", C.c_void_p), ("position_offset", C.c_int), ("qstart", C.c_int),
                ("qend", C.c_int), ("score_token", C.c_float), ("identity", C.c_float), ("nm", C.c_int)]


def library_path():
    return os.path.join(_HERE, "libngm_hip.so")


_lib = None


def load_library():
    """Loads the HIP library; raises loudly if it has not been built (no fallback of any kind)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch wheels bundle their own HIP/HSA runtime (same SONAME as /opt/rocm's).  Two HIP runtimes in
    # one process cannot both own the GPU, so if torch is going to be used in this process (bench.py,
    # the device-resident tests) it must be loaded FIRST; this library then binds to that runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = library_path()
    if not os.path.exists(path):
        raise NgmHipError("HIP extension missing: %s (run `python -m nextgenmap_amd.build`)" % path)
    lib = C.CDLL(path)
    lib.ngm_hip_create.restype = C.c_void_p
    lib.ngm_hip_create.argtypes = [C.c_int, C.POINTER(Params)]
    lib.ngm_hip_destroy.argtypes = [C.c_void_p]
    lib.ngm_hip_last_error.restype = C.c_char_p
    lib.ngm_hip_last_error.argtypes = [C.c_void_p]
    lib.ngm_hip_device_count.restype = C.c_int
    lib.ngm_hip_score_batch_size.argtypes = [C.c_void_p]
    lib.ngm_hip_align_batch_size.argtypes = [C.c_void_p]
    lib.ngm_hip_batch_score.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ngm_hip_batch_align.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ngm_hip_score_device.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ngm_hip_align_run_stride.argtypes = [C.c_void_p]
    lib.ngm_hip_align_device.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_void_p]
    lib.ngm_hip_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.ngm_hip_set_profiling.argtypes = [C.c_void_p, C.c_int]
    _lib = lib
    return lib


def _ptr(x):
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    return int(x)


class Engine:
    """One IAlignment instance (NGM creates one per CS thread, src/CS.cpp:455-461)."""

    def __init__(self, qry_max_len, corridor, match=10, mismatch=15, gap_read=20, gap_ref=20, device=0,
                 variant=VARIANT_OCL_GPU, hard_clip=0, silent_clip=0, max_batch=0, personality=PERSONALITY_LINEAR,
                 gap_extend=0, alt_scoring=0, match_bonus_tt=0, match_bonus_tc=0, alt_cigar=None):
        self.lib = load_library()
        self.q, self.c = int(qry_max_len), int(corridor)
        self.personality = int(personality)
        p = Params(ABI_VERSION, self.q, self.c, match, mismatch, gap_read, gap_ref, variant, hard_clip, silent_clip, max_batch,
                   self.personality, gap_extend, alt_scoring, match_bonus_tt, match_bonus_tc, alt_scoring if alt_cigar is None else alt_cigar)
        self.h = self.lib.ngm_hip_create(device, C.byref(p))
        if not self.h:
            raise NgmHipError(self.lib.ngm_hip_last_error(None).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.ngm_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, r, n):
        if r < 0:
            raise NgmHipError(self.lib.ngm_hip_last_error(self.h).decode())
        return r

    # ---- IAlignment surface ------------------------------------------------------------------
    def GetScoreBatchSize(self):
        return self.lib.ngm_hip_score_batch_size(self.h)

    def GetAlignBatchSize(self):
        return self.lib.ngm_hip_align_batch_size(self.h)

    def _ptr_lists(self, ref, qry):
        ref = np.ascontiguousarray(ref, dtype=np.uint8)
        qry = np.ascontiguousarray(qry, dtype=np.uint8)
        n = qry.shape[0]
        assert qry.shape[1] >= self.q and ref.shape[1] >= self.q + self.c and ref.shape[0] == n
        rp = (C.c_void_p * n)(*[ref.ctypes.data + i * ref.strides[0] for i in range(n)])
        qp = (C.c_void_p * n)(*[qry.ctypes.data + i * qry.strides[0] for i in range(n)])
        return n, ref, qry, rp, qp

    def BatchScore(self, mode, ref, qry, dirs=None):
        """ref [n, >=q+c] uint8 rows, qry [n, >=q] uint8 rows (NUL padded) -> float32[n].  dirs: extData (alt_scoring)."""
        n, ref, qry, rp, qp = self._ptr_lists(ref, qry)
        out = np.full(n, -1.0, dtype=np.float32)  # ScoreBuffer.cpp:120
        d = None if dirs is None else np.ascontiguousarray(dirs, dtype=np.uint8)
        r = self._check(self.lib.ngm_hip_batch_score(self.h, mode, n, rp, qp, out.ctypes.data, None if d is None else d.ctypes.data), n)
        assert r == n
        return out

    def BatchAlign(self, mode, ref, qry, dirs=None):
        """-> list of dicts(cigar, md, position_offset, qstart, qend, score_token, identity, nm)."""
        n, ref, qry, rp, qp = self._ptr_lists(ref, qry)
        stride = 4 * max(1, self.q)  # AlignmentBuffer.cpp:106-109
        cig = np.zeros((n, stride), dtype=np.uint8)
        md = np.zeros((n, stride), dtype=np.uint8)
        cig[:, :3] = ord("!")
        md[:, :3] = ord("!")
        outs = (AlignOut * n)()
        for i in range(n):
            outs[i].cigar = cig.ctypes.data + i * stride
            outs[i].md = md.ctypes.data + i * stride
        d = None if dirs is None else np.ascontiguousarray(dirs, dtype=np.uint8)
        r = self._check(self.lib.ngm_hip_batch_align(self.h, mode, n, rp, qp, outs, None if d is None else d.ctypes.data), n)
        assert r == n
        res = []
        for i in range(n):
            o = outs[i]
            res.append(dict(cigar=bytes(cig[i]).split(b"\0", 1)[0], md=bytes(md[i]).split(b"\0", 1)[0],
                            position_offset=o.position_offset, qstart=o.qstart, qend=o.qend,
                            score_token=o.score_token, identity=o.identity, nm=o.nm))
        return res

    # ---- HBM-resident batches ----------------------------------------------------------------
    def set_profiling(self, on):
        self.lib.ngm_hip_set_profiling(self.h, 1 if on else 0)

    def score_device(self, mode, n, d_ref, d_qry, d_scores, stream=None):
        return self._check(self.lib.ngm_hip_score_device(self.h, mode, n, _ptr(d_ref), _ptr(d_qry), _ptr(d_scores),
                                                          _ptr(stream)), n)

    def align_run_stride(self):
        return self.lib.ngm_hip_align_run_stride(self.h)

    def align_device(self, mode, n, d_ref, d_qry, d_records, d_runs, run_stride, stream=None):
        return self._check(self.lib.ngm_hip_align_device(self.h, mode, n, _ptr(d_ref), _ptr(d_qry), _ptr(d_records),
                                                          _ptr(d_runs), run_stride, _ptr(stream)), n)

    def last_kernel_ms(self):
        ms = (C.c_float * 3)()
        self._check(self.lib.ngm_hip_last_kernel_ms(self.h, ms), 0)
        return [ms[0], ms[1], ms[2]]
