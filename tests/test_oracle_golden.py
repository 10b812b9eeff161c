"""The C restatement (oracle/) against golden vectors captured from the REFERENCE'S OWN KERNELS run
on the MI355X (oracle/make_goldens.py; NextGenMap's OpenCL kernels compiled unmodified for gfx950).
This is what pins the oracle; it runs on CPU."""
import glob
import os

import numpy as np
import pytest

import oracle_lib as O

GOLDEN = [p for p in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ngm_ocl_*.npz"))) if not p.endswith("_cigar.npz")]


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
    scoring, dirs, _ = O.golden_case(g)
    sc = O.oracle_score(mode, ref, qry, c, scoring, variant=variant, dirs=dirs)
    assert np.array_equal(sc, g[mn + "_score"])
    # BatchAlign kernels (oclSW_Score[Global] + oclSW_Backtracking): raw outputs, bit-exact
    res, rle, valid, _ = O.oracle_trace(mode, ref, qry, c, scoring, variant=variant, dirs=dirs)
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


CIGAR_GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ngm_ocl_*_cigar.npz")))


def cigar_golden_rows(path, mode, clip):
    """(row indices, expected tuples) from the fixtures written by oracle/make_cigar_goldens.py: the output of the
    REFERENCE'S OWN SWOclCigar::computeCigarMD (lib/mason/opencl/SWOclCigar.cpp:430-615) on the golden RLE rows."""
    g = np.load(path)
    mn = "local" if mode == 0 else "endfree"
    cn = ("soft", "hard", "silent")[clip]
    rows = g[mn + "_rows"]
    f = lambda k: g["%s_%s_%s" % (mn, cn, k)]
    want = [(bool(f("ok")[j]), bytes(f("cigar")[j]), bytes(f("md")[j]), int(f("nm")[j]), np.float32(f("identity")[j]).tobytes(),
             int(f("qstart")[j]), int(f("qend")[j]), int(f("position_offset")[j]), float(f("score_token")[j])) for j in range(len(rows))]
    return rows, want


def test_cigar_goldens_present():
    assert len(CIGAR_GOLDEN) >= 7


@pytest.mark.parametrize("path", CIGAR_GOLDEN, ids=[os.path.basename(p) for p in CIGAR_GOLDEN])
@pytest.mark.parametrize("mode", [0, 1], ids=["local", "endfree"])
@pytest.mark.parametrize("clip", [0, 1, 2], ids=["soft", "hardclip", "silentclip"])
def test_oracle_cigar_md_matches_reference_function(path, mode, clip):
    """CIGAR / MD / NM / Identity / QStart / QEnd of the restatement == the reference's computeCigarMD."""
    g = np.load(path.replace("_cigar.npz", ".npz"))
    ref, qry, c, variant = g["ref"], g["qry"], int(g["c"]), int(g["variant"])
    if clip and qry.shape[1] > 152:
        pytest.skip("clipping modes only change the string conversion: covered on the shapes up to 150 bp (CPU-suite time)")
    rows, want = cigar_golden_rows(path, mode, clip)
    scoring, dirs, _ = O.golden_case(g)
    res, cig, md = O.oracle_align(mode, ref, qry, c, scoring, variant=variant, hard_clip=int(clip == 1), silent_clip=int(clip == 2), nthreads=8, dirs=dirs)
    for j, i in enumerate(rows):
        have = (bool(res["ok"][i]), cig[i], md[i], int(res["nm"][i]), np.float32(res["identity"][i]).tobytes(), int(res["qstart"][i]),
                int(res["qend"][i]), int(res["position_offset"][i]), float(res["score_token"][i]))
        assert have == want[j], "row %d" % i
