"""The C restatement (oracle/) against golden vectors captured from the REFERENCE'S OWN KERNELS run
on the MI355X (oracle/make_goldens.py; NextGenMap's OpenCL kernels compiled unmodified for gfx950).
This is what pins the oracle; it runs on CPU."""
import glob
import os

import numpy as np
import pytest

import oracle_lib as O

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ngm_ocl_*.npz")))


def test_goldens_present():
    assert len(GOLDEN) >= 7


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
@pytest.mark.parametrize("mode", [0, 1], ids=["local", "endfree"])
def test_oracle_matches_reference_kernels(path, mode):
    g = np.load(path)
    ref, qry, c, variant = g["ref"], g["qry"], int(g["c"]), int(g["variant"])
    q = qry.shape[1]
    al = 2 * q + c + 1
    mn = "local" if mode == 0 else "endfree"
    # BatchScore kernels (oclSW / oclSW_Global): bit-exact
    sc = O.oracle_score(mode, ref, qry, c, variant=variant)
    assert np.array_equal(sc, g[mn + "_score"])
    # BatchAlign kernels (oclSW_Score[Global] + oclSW_Backtracking): raw outputs, bit-exact
    res, rle, valid, _ = O.oracle_trace(mode, ref, qry, c, variant=variant)
    assert np.array_equal(valid, g[mn + "_valid"])
    assert np.array_equal(res[:, :3], g[mn + "_res"][:, :3])
    assert np.array_equal(res[valid, 3], g[mn + "_res"][valid, 3])
    gr = g[mn + "_rle"]
    for i in np.nonzero(valid)[0]:
        off = int(res[i, 3])
        assert np.array_equal(rle[i, off:al], gr[i, off:al]), "rle row %d" % i


def test_survey_known_answer():
    """100 bp read, 1 mismatch, 2 bp deletion, window 122 -> oclSW 935,
    '0S 30= 1X 19= 2D 50= 0S', refpos 10 (values measured on the reference while surveying)."""
    rng = np.random.default_rng(5)
    q, c = 102, 20
    win = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, q + c)].copy()
    read = np.concatenate([win[10:60], win[62:112]]).copy()
    read[30] = ord("A") if read[30] != ord("A") else ord("C")
    qry = np.zeros((1, q), np.uint8)
    qry[0, :100] = read
    ref = win[None, :]
    assert O.oracle_score(0, ref, qry, c)[0] == 935.0
    res, cig, md = O.oracle_align(0, ref, qry, c)
    assert cig[0] == b"50M2D50M" and res["position_offset"][0] == 10 and res["nm"][0] == 3
    assert md[0].startswith(b"30") and b"^" in md[0]


def test_threads_agree():
    from pairgen import make_pairs
    ref, qry = make_pairs(300, 52, 12, seed=7, read_len=50)
    a = O.oracle_score(0, ref, qry, 12, nthreads=1)
    b = O.oracle_score(0, ref, qry, 12, nthreads=4)
    assert np.array_equal(a, b)
