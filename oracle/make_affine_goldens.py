#!/usr/bin/env python3
"""Capture golden vectors of the affine personality from the REFERENCE's own EndToEndAffine (SeqAn), built by
oracle/ngm_ref.mk as oracle/_ref/ngm/ngm_affine_ref.  Runs in the build container (no GPU needed).
Output: tests/golden/ngm_affine_seqan.npz (inputs + every output field)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from pairgen import make_pairs  # noqa: E402

q, c, n = 102, 20, 1500
ref, qry = make_pairs(n, q, c, seed=4242, read_len=100)
out = dict(ref=ref, qry=qry, q=q, c=c)
for mode, mn in ((0, "local"), (1, "endfree")):
    want = O.reference_affine(mode, ref, qry, c)
    valid = np.array([w is not None for w in want])
    out[mn + "_valid"] = valid
    out[mn + "_score"] = np.array([w[0] if w else 0 for w in want], np.float32)
    for k, j in (("position_offset", 2), ("qstart", 3), ("qend", 4), ("nm", 5)):
        out[mn + "_" + k] = np.array([w[j] if w else 0 for w in want], np.int32)
    out[mn + "_identity"] = np.array([w[6] if w else 0 for w in want], np.float32)
    out[mn + "_cigars"] = np.frombuffer(b"\n".join(w[1] if w else b"" for w in want), np.uint8)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ngm_affine_seqan.npz"), **out)
print("wrote", n, "pairs x 2 modes")
