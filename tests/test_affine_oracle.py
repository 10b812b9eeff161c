"""The affine (SeqAn) restatement against the REFERENCE'S OWN EndToEndAffine compiled from /root/reference
(oracle/_ref/ngm/ngm_affine_ref, built by oracle/ngm_ref.mk; driver oracle/affine_ref_main.cpp), plus committed
golden vectors captured from it so the check also runs where the reference tree is absent."""
import os

import numpy as np
import pytest

import oracle_lib as O
from pairgen import make_pairs

HAVE_REF = os.path.exists(O.AFFINE_REF)
GOLD = os.path.join(os.path.dirname(__file__), "golden", "ngm_affine_seqan.npz")


def _cmp(want, sc, res, cig):
    bad = []
    for i, w in enumerate(want):
        if w is None:
            continue
        got = (int(sc[i]), cig[i], int(res["position_offset"][i]), int(res["qstart"][i]), int(res["qend"][i]), int(res["nm"][i]))
        idok = (np.float32(res["identity"][i]) == w[6]) or (np.isnan(res["identity"][i]) and np.isnan(w[6]))
        if got != tuple(w[:6]) or not idok:
            bad.append((i, got, w))
    return bad


@pytest.mark.skipif(not HAVE_REF, reason="affine reference harness not built (oracle/ngm_ref.mk)")
@pytest.mark.parametrize("q,c,rl,n", [(32, 8, 28, 1500), (102, 20, 100, 1200), (152, 27, 150, 800), (252, 80, 250, 200)])
@pytest.mark.parametrize("mode", [0, 1], ids=["local", "endfree"])
def test_restatement_matches_reference_seqan(q, c, rl, n, mode):
    ref, qry = make_pairs(n, q, c, seed=40 + q + mode, read_len=rl)
    want = O.reference_affine(mode, ref, qry, c)
    sc, res, cig = O.oracle_affine(mode, ref, qry, c, nthreads=4)
    bad = _cmp(want, sc, res, cig)
    assert not bad, bad[:3]


@pytest.mark.skipif(not HAVE_REF, reason="affine reference harness not built (oracle/ngm_ref.mk)")
def test_custom_scoring_matches_reference_seqan():
    ref, qry = make_pairs(800, 102, 20, seed=77, read_len=100)
    scoring = dict(match=7, mismatch=-11, gap_open=-19, gap_extend=-2)
    for mode in (0, 1):
        want = O.reference_affine(mode, ref, qry, 20, scoring)
        sc, res, cig = O.oracle_affine(mode, ref, qry, 20, scoring)
        assert not _cmp(want, sc, res, cig)


def test_restatement_matches_committed_goldens():
    g = np.load(GOLD, allow_pickle=False)
    for mode, mn in ((0, "local"), (1, "endfree")):
        ref, qry, c = g["ref"], g["qry"], int(g["c"])
        sc, res, cig = O.oracle_affine(mode, ref, qry, c)
        valid = g[mn + "_valid"]
        assert np.array_equal(sc[valid], g[mn + "_score"][valid])
        for k in ("position_offset", "qstart", "qend", "nm"):
            assert np.array_equal(res[k][valid], g[mn + "_" + k][valid]), k
        assert np.array_equal(res["identity"][valid].view(np.uint32), g[mn + "_identity"][valid].view(np.uint32))
        want_cig = bytes(g[mn + "_cigars"]).split(b"\n")
        assert [cig[i] for i in np.nonzero(valid)[0]] == [want_cig[i] for i in np.nonzero(valid)[0]]
