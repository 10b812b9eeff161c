"""Readers for the cache files the REAL reference program writes next to a FASTA
(<ref>-enc.2.ngm: src/SequenceProvider.cpp:189-208, <ref>-ht-<k>-<skip>.3.ngm: src/PrefixTable.cpp:819-855),
plus a runner for the reference binary built by oracle/ngm_ref.mk.  Test infrastructure."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NGM_CORE = os.path.join(ROOT, "oracle", "_ref", "ngm", "ngm-core")
NGM_CORE_DEBUG = os.path.join(ROOT, "oracle", "_ref", "ngm", "ngm-core-debug")


def have_reference_binary():
    return os.path.exists(NGM_CORE) and os.path.exists(NGM_CORE_DEBUG)


def run_ngm(args, debug=False, cwd=None, timeout=1800):
    exe = NGM_CORE_DEBUG if debug else NGM_CORE
    return subprocess.run([exe] + list(args), capture_output=True, text=True, cwd=cwd, timeout=timeout)


def read_ht_file(path):
    """-> dict(k, skip, counts[4^k] (as a lookup sees them), raw_counts, positions grouped by k-mer)"""
    with open(path, "rb") as f:
        cookie, k, skip, units, index_size = np.fromfile(f, np.uint32, 5)
        assert units == 1
        table_len = int(np.fromfile(f, np.uint32, 1)[0])
        idx = np.fromfile(f, np.dtype([("tab", "<u4"), ("rev", "i1")]), int(index_size))
        pos = np.fromfile(f, np.uint32, table_len)
    nk = 4 ** int(k)
    tab = idx["tab"].astype(np.int64)
    raw = (tab[1:nk + 1] - tab[:nk]).astype(np.uint32)
    used = idx["rev"][:nk] != 0
    counts = np.where(used, raw, 0).astype(np.uint32)
    return dict(k=int(k), skip=int(skip), counts=counts, raw_counts=raw, starts=(tab[:nk] - 1), positions=pos)


def read_enc_file(path):
    with open(path, "rb") as f:
        cookie, ref_count = np.fromfile(f, np.uint32, 2)
        bin_ref_index, enc_size = np.fromfile(f, np.uint64, 2)
        refidx = np.fromfile(f, np.dtype([("SeqId", "<u4"), ("Flags", "<u4"), ("SeqStart", "<u8"), ("SeqLen", "<u4"),
                                          ("NameLen", "<u4"), ("name", "S100"), ("pad", "V4")]), int(ref_count))
        data = np.fromfile(f, np.uint8, int(enc_size))
    return dict(n_bases=int(bin_ref_index), refidx=refidx, data=data)
