"""ctypes mirror of include/ngm_pipeline.h: HBM-resident reference + k-mer index, candidate search and the
single-end mapping path.  Used by the tests, bench.py and the ngm-style command line (nextgenmap_amd.cli)."""
import ctypes as C

import numpy as np

from .engine import NgmHipError, load_library


class RefParams(C.Structure):
    _fields_ = [("kmer", C.c_int), ("kmer_skip", C.c_int), ("bin_size", C.c_int)]


class MapperParams(C.Structure):
    _fields_ = [("qry_max_len", C.c_int), ("corridor", C.c_int), ("match_bonus", C.c_int), ("mismatch_penalty", C.c_int),
                ("gap_read_penalty", C.c_int), ("gap_ref_penalty", C.c_int), ("mode", C.c_int), ("variant", C.c_int),
                ("sensitivity", C.c_float), ("kmer_min", C.c_float), ("max_cmrs", C.c_int), ("max_kfreq", C.c_int),
                ("hard_clip", C.c_int), ("silent_clip", C.c_int), ("personality", C.c_int), ("gap_extend_penalty", C.c_int),
                ("min_insert_size", C.c_int), ("max_insert_size", C.c_int), ("pair_score_cutoff", C.c_float),
                ("topn", C.c_int), ("strata", C.c_int), ("bs_mapping", C.c_int), ("bs_cutoff", C.c_int), ("bs_read_skip", C.c_int),
                ("match_bonus_tt", C.c_int), ("match_bonus_tc", C.c_int), ("slam_seq", C.c_int)]


HIT_DTYPE = np.dtype([("mapped", "i4"), ("contig", "i4"), ("pos", "u8"), ("reverse", "i4"), ("mapq", "i4"),
                      ("score", "f4"), ("identity", "f4"), ("nm", "i4"), ("qstart", "i4"), ("qend", "i4"),
                      ("n_candidates", "i4"), ("n_best", "i4"), ("max_votes", "f4"), ("pair_flags", "i4")], align=True)

_bound = False


def _lib():
    global _bound
    lib = load_library()
    if not _bound:
        lib.ngm_pipeline_last_error.restype = C.c_char_p
        lib.ngm_ref_create.restype = C.c_void_p
        lib.ngm_ref_create.argtypes = [C.c_int, C.POINTER(RefParams), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.ngm_ref_create_from_fasta.restype = C.c_void_p
        lib.ngm_ref_create_from_fasta.argtypes = [C.c_int, C.POINTER(RefParams), C.c_char_p]
        lib.ngm_ref_create_from_cache.restype = C.c_void_p
        lib.ngm_ref_create_from_cache.argtypes = [C.c_int, C.POINTER(RefParams), C.c_char_p]
        lib.ngm_ref_destroy.argtypes = [C.c_void_p]
        lib.ngm_ref_contig_count.argtypes = [C.c_void_p]
        lib.ngm_ref_contig_name.restype = C.c_char_p
        lib.ngm_ref_contig_name.argtypes = [C.c_void_p, C.c_int]
        for f in ("ngm_ref_contig_start", "ngm_ref_contig_len"):
            getattr(lib, f).restype = C.c_uint64
            getattr(lib, f).argtypes = [C.c_void_p, C.c_int]
        for f in ("ngm_ref_concat_len", "ngm_ref_index_entries"):
            getattr(lib, f).restype = C.c_uint64
            getattr(lib, f).argtypes = [C.c_void_p]
        lib.ngm_ref_auto_max_kfreq.argtypes = [C.c_void_p]
        lib.ngm_ref_index_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        if not hasattr(lib, "ngm_mapper_create"):
            raise NgmHipError("libngm_hip.so was built without the mapping pipeline")
        lib.ngm_ref_write_ngm_cache.argtypes = [C.c_void_p, C.c_char_p]
        lib.ngm_ref_decode.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
        lib.ngm_ref_convert.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
        lib.ngm_mapper_create.restype = C.c_void_p
        lib.ngm_mapper_create.argtypes = [C.c_void_p, C.POINTER(MapperParams)]
        lib.ngm_mapper_destroy.argtypes = [C.c_void_p]
        lib.ngm_mapper_cs.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.ngm_mapper_cs_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.ngm_mapper_map_se.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.ngm_mapper_map_se_resident.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.ngm_mapper_map_pe_resident.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.ngm_mapper_cs_counters.argtypes = [C.c_void_p, C.c_void_p]
        lib.ngm_mapper_path_counters.argtypes = [C.c_void_p, C.c_void_p]
        lib.ngm_mapper_order_table_reads.argtypes = [C.c_void_p, C.c_void_p]
        lib.ngm_mapper_heavy_counters.argtypes = [C.c_void_p, C.c_void_p]
        lib.ngm_mapper_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        lib.ngm_mapper_last_order_replay_ms.restype = C.c_float
        lib.ngm_mapper_last_order_replay_ms.argtypes = [C.c_void_p]
        lib.ngm_bgzf_create.restype = C.c_void_p
        lib.ngm_bgzf_create.argtypes = [C.c_int]
        lib.ngm_bgzf_destroy.argtypes = [C.c_void_p]
        lib.ngm_bgzf_bound.restype = C.c_size_t
        lib.ngm_bgzf_bound.argtypes = [C.c_size_t]
        lib.ngm_bgzf_compress.restype = C.c_longlong
        lib.ngm_bgzf_compress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        lib.ngm_bgzf_last_kernel_ms.restype = C.c_float
        lib.ngm_bgzf_last_kernel_ms.argtypes = [C.c_void_p]
        _bound = True
    return lib


def _err():
    return NgmHipError(_lib().ngm_pipeline_last_error().decode())


class Bgzf:
    """BGZF blocks written by the GPU (include/ngm_pipeline.h, ngm_bgzf_*): `ngm --bam`'s block compressor."""

    def __init__(self, device=0):
        self._h = _lib().ngm_bgzf_create(device)
        if not self._h:
            raise _err()

    def compress(self, data):
        lib = _lib()
        data = bytes(data)
        cap = lib.ngm_bgzf_bound(len(data))
        out = C.create_string_buffer(cap)
        n = lib.ngm_bgzf_compress(self._h, data, len(data), out, cap)
        if n < 0:
            raise _err()
        return out.raw[:n]

    def last_kernel_ms(self):
        return float(_lib().ngm_bgzf_last_kernel_ms(self._h))

    def close(self):
        if self._h:
            _lib().ngm_bgzf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Reference:
    """Encoded reference + k-mer index resident in HBM (SequenceProvider + CompactPrefixTable of NGM)."""

    def __init__(self, handle, kmer):
        self.lib = _lib()
        self.h = handle
        self.kmer = kmer

    @classmethod
    def from_contigs(cls, contigs, names=None, device=0, kmer=13, kmer_skip=2, bin_size=2):
        lib = _lib()
        n = len(contigs)
        arrs = [np.ascontiguousarray(c, dtype=np.uint8) for c in contigs]
        names = [("chr%d" % (i + 1)) if names is None else names[i] for i in range(n)]
        cn = (C.c_char_p * n)(*[s.encode() for s in names])
        cs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        cl = (C.c_uint64 * n)(*[a.size for a in arrs])
        p = RefParams(kmer, kmer_skip, bin_size)
        h = lib.ngm_ref_create(device, C.byref(p), n, cn, cs, cl)
        if not h:
            raise _err()
        return cls(h, kmer)

    @classmethod
    def from_fasta(cls, path, device=0, kmer=13, kmer_skip=2, bin_size=2):
        lib = _lib()
        p = RefParams(kmer, kmer_skip, bin_size)
        h = lib.ngm_ref_create_from_fasta(device, C.byref(p), path.encode())
        if not h:
            raise _err()
        return cls(h, kmer)

    @classmethod
    def from_cache(cls, fasta_path, device=0, kmer=13, kmer_skip=2, bin_size=2):
        """NextGenMap's own <fasta>-enc.2.ngm / <fasta>-ht-<k>-<skip>.3.ngm cache files."""
        lib = _lib()
        p = RefParams(kmer, kmer_skip, bin_size)
        h = lib.ngm_ref_create_from_cache(device, C.byref(p), fasta_path.encode())
        if not h:
            raise _err()
        return cls(h, kmer)

    def close(self):
        if getattr(self, "h", None):
            self.lib.ngm_ref_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def contigs(self):
        return [(self.lib.ngm_ref_contig_name(self.h, i).decode(), int(self.lib.ngm_ref_contig_start(self.h, i)),
                 int(self.lib.ngm_ref_contig_len(self.h, i))) for i in range(self.lib.ngm_ref_contig_count(self.h))]

    @property
    def concat_len(self):
        return int(self.lib.ngm_ref_concat_len(self.h))

    @property
    def auto_max_kfreq(self):
        return self.lib.ngm_ref_auto_max_kfreq(self.h)

    @property
    def index_entries(self):
        return int(self.lib.ngm_ref_index_entries(self.h))

    def index_copy(self):
        nk = 1 << (2 * self.kmer)
        counts = np.zeros(nk, np.uint32)
        raw = np.zeros(nk, np.uint32)
        pos = np.zeros(max(1, self.index_entries), np.uint32)
        if self.lib.ngm_ref_index_copy(self.h, counts.ctypes.data, raw.ctypes.data, pos.ctypes.data) < 0:
            raise _err()
        return counts, raw, pos[:self.index_entries]

    def write_ngm_cache(self, fasta_path):
        """Write NextGenMap's own index/genome cache files next to fasta_path (the reference program then loads
        them instead of rebuilding)."""
        if self.lib.ngm_ref_write_ngm_cache(self.h, fasta_path.encode()) < 0:
            raise _err()

    def decode(self, offset, buffer_len):
        out = np.zeros(buffer_len, np.uint8)
        r = self.lib.ngm_ref_decode(self.h, offset, buffer_len, out.ctypes.data)
        if r < 0:
            raise _err()
        return bool(r), bytes(out)

    def convert(self, pos):
        c, p = C.c_int(0), C.c_uint64(0)
        ok = self.lib.ngm_ref_convert(self.h, pos, C.byref(c), C.byref(p))
        return (c.value, p.value) if ok else None


class Mapper:
    """CS -> gather -> score -> top-1 selection -> align for single-end reads (one CS thread's worth of NGM)."""

    def __init__(self, ref, qry_max_len, corridor, sensitivity=0.5, match=10, mismatch=15, gap_read=20, gap_ref=20,
                 mode=0, variant=0, kmer_min=0.0, max_cmrs=2 ** 31 - 1, max_kfreq=0, hard_clip=0, silent_clip=0, personality=0,
                 gap_extend=0, min_insert_size=0, max_insert_size=1000, pair_score_cutoff=0.9, topn=1,
                 strata=0, bs_mapping=0, bs_cutoff=6, bs_read_skip=2, match_bonus_tt=4, match_bonus_tc=4, slam_seq=0):
        self.lib = _lib()
        self.ref = ref
        self.q, self.c = qry_max_len, corridor
        p = MapperParams(qry_max_len, corridor, match, mismatch, gap_read, gap_ref, mode, variant, sensitivity, kmer_min,
                         max_cmrs, max_kfreq, hard_clip, silent_clip, personality, gap_extend, min_insert_size, max_insert_size,
                         pair_score_cutoff, topn, strata, bs_mapping, bs_cutoff, bs_read_skip, match_bonus_tt, match_bonus_tc, slam_seq)
        self.topn = max(1, topn)
        self.h = self.lib.ngm_mapper_create(ref.h, C.byref(p))
        if not self.h:
            raise _err()

    def close(self):
        if getattr(self, "h", None):
            self.lib.ngm_mapper_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def reads_to_rows(reads, q):
        """list of uint8 arrays / bytes -> [n, q] NUL padded, upper-cased, non-ACGT -> N, truncated to q-1
        (IParser.h:59-121)."""
        out = np.zeros((len(reads), q), np.uint8)
        lut = np.full(256, ord("N"), np.uint8)
        for a, b in zip(b"ACGTacgt", b"ACGTACGT"):
            lut[a] = b
        for i, r in enumerate(reads):
            a = np.frombuffer(bytes(r), dtype=np.uint8)[:q - 1]
            out[i, :a.size] = lut[a]
        return out

    def candidate_search(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        n = rows.shape[0]
        offs = np.zeros(n + 1, np.uint32)
        mx = np.zeros(n, np.float32)
        if self.lib.ngm_mapper_cs(self.h, n, rows.ctypes.data, offs.ctypes.data, mx.ctypes.data) < 0:
            raise _err()
        tot = int(offs[-1])
        loc = np.zeros(max(1, tot), np.uint64)
        strand = np.zeros(max(1, tot), np.uint8)
        votes = np.zeros(max(1, tot), np.float32)
        if self.lib.ngm_mapper_cs_fetch(self.h, loc.ctypes.data, strand.ctypes.data, votes.ctypes.data) < 0:
            raise _err()
        return offs, mx, loc[:tot], strand[:tot], votes[:tot]

    def map_pe_raw(self, rows, d_rows=None, out=None):
        """Paired-end: rows 2i and 2i+1 are mates.  Same outputs as map_se_raw."""
        return self.map_se_raw(rows, d_rows, out, paired=True)

    def map_pe(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        hits, cig, md = self.map_pe_raw(rows)
        return hits, [bytes(r).split(b"\0", 1)[0] for r in cig], [bytes(r).split(b"\0", 1)[0] for r in md]

    def map_se_raw(self, rows, d_rows=None, out=None, paired=False):
        """rows: [n, q] uint8 host array; d_rows: optional device copy (torch tensor / pointer).  Returns
        (hits, cigar bytes [n, 4q], md bytes [n, 4q]) without turning the strings into Python objects."""
        n = rows.shape[0]
        stride = 4 * max(1, self.q)
        if out is None:
            no = n * (1 if paired else self.topn)
            out = (np.zeros(no, HIT_DTYPE), np.zeros((no, stride), np.uint8), np.zeros((no, stride), np.uint8))
        hits, cig, md = out
        dp = None if d_rows is None else (d_rows.data_ptr() if hasattr(d_rows, "data_ptr") else int(d_rows))
        fn = self.lib.ngm_mapper_map_pe_resident if paired else self.lib.ngm_mapper_map_se_resident
        r = fn(self.h, n, rows.ctypes.data, dp, hits.ctypes.data, cig.ctypes.data, md.ctypes.data)
        if r < 0:
            raise _err()
        return hits, cig, md

    def map_se(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        hits, cig, md = self.map_se_raw(rows)
        return hits, [bytes(r).split(b"\0", 1)[0] for r in cig], [bytes(r).split(b"\0", 1)[0] for r in md]

    def last_kernel_ms(self):
        ms = (C.c_float * 8)()
        self.lib.ngm_mapper_last_kernel_ms(self.h, ms)
        return list(ms)

    def last_order_replay_ms(self):
        """GPU time of the candidate-order replays of the last call (their own stream)"""
        return float(self.lib.ngm_mapper_last_order_replay_ms(self.h))

    def path_counters(self):
        """summed over all batches: reads searched, candidates, reads re-run by the exact search (LDS table / global-memory table),
        reads whose candidate order was replayed, of those by the exact global-memory replay, reads left with an undetermined order"""
        out = np.zeros(8, np.uint64)
        self.lib.ngm_mapper_path_counters(self.h, out.ctypes.data)
        return dict(zip(("reads", "candidates", "exact_lds", "exact_global", "order_replayed", "order_exact_global", "order_undetermined", "heavy"), (int(x) for x in out[:8])))

    def heavy_counters(self):
        """of path_counters()["heavy"]: second passes, table passes started over, reads sent on to the exact kernels, regrown table pools"""
        out = np.zeros(4, np.uint64)
        self.lib.ngm_mapper_heavy_counters(self.h, out.ctypes.data)
        return dict(zip(("second_passes", "table_pass_restarts", "sent_on", "pool_regrown"), (int(x) for x in out)))

    def order_table_reads(self):
        """of path_counters()["order_exact_global"]: the reads the bucket replay left to the replay with a table in global memory"""
        out = np.zeros(1, np.uint64)
        self.lib.ngm_mapper_order_table_reads(self.h, out.ctypes.data)
        return int(out[0])

    def cs_counters(self):
        """(k-mers looked up, index hits voted, candidates) of the last candidate search."""
        out = np.zeros(3, np.uint64)
        self.lib.ngm_mapper_cs_counters(self.h, out.ctypes.data)
        return [int(x) for x in out]
