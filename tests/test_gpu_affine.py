"""GPU parity of the affine personality (`ngm --affine`): the HIP kernels behind the C ABI against the CPU
restatement of EndToEndAffine / SeqAn (oracle/ngm_affine_oracle.c, itself pinned on the reference's own code)
and against the golden vectors captured from the reference's own EndToEndAffine."""
import os

import numpy as np
import pytest

import oracle_lib as O
from pairgen import make_pairs

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "ngm_affine_seqan.npz")


def _engine(q, c, **kw):
    import nextgenmap_amd as N
    from nextgenmap_amd import engine as E
    kw.setdefault("gap_read", 33)
    kw.setdefault("gap_ref", 33)
    kw.setdefault("gap_extend", 3)
    return N.Engine(q, c, personality=E.PERSONALITY_AFFINE, **kw)


def _check(eng, mode, ref, qry, c, scoring=None):
    sc, res, cig = O.oracle_affine(mode, ref, qry, c, scoring, nthreads=8)
    got = eng.BatchScore(mode, ref, qry)
    bad = np.nonzero(got != sc)[0]
    assert bad.size == 0, (bad[:5], got[bad[:5]], sc[bad[:5]])
    al = eng.BatchAlign(mode, ref, qry)
    for i, a in enumerate(al):
        want = (cig[i], int(res["position_offset"][i]), int(res["qstart"][i]), int(res["qend"][i]), int(res["nm"][i]))
        have = (a["cigar"], a["position_offset"], a["qstart"], a["qend"], a["nm"])
        assert have == want, (i, have, want, bytes(ref[i]), bytes(qry[i]))
        wi, hi = np.float32(res["identity"][i]), np.float32(a["identity"])
        assert wi == hi or (np.isnan(wi) and np.isnan(hi)), (i, wi, hi)
        assert a["md"] == b"!!!"  # EndToEndAffine leaves pBuffer2 alone


@pytest.mark.parametrize("q,c,rl,n", [(32, 8, 28, 3000), (102, 20, 100, 3000), (152, 27, 150, 3000), (252, 42, 250, 1200),
                                      (252, 80, 250, 400), (62, 12, 60, 1000), (128, 19, 120, 1000)])
@pytest.mark.parametrize("mode", [0, 1], ids=["local", "endfree"])
def test_affine_matches_oracle(q, c, rl, n, mode):
    ref, qry = make_pairs(n, q, c, seed=500 + q + c + mode, read_len=rl)
    eng = _engine(q, c)
    _check(eng, mode, ref, qry, c)
    eng.close()


@pytest.mark.parametrize("mode", [0, 1], ids=["local", "endfree"])
def test_affine_indel_rich_pairs(mode):
    q, c = 152, 27
    ref, qry = make_pairs(4000, q, c, seed=91 + mode, read_len=150, sub_rate=0.05, indel_rate=0.03, mix=(0.9, 0.05, 0.05))
    eng = _engine(q, c)
    _check(eng, mode, ref, qry, c)
    eng.close()


def test_affine_custom_scoring():
    q, c = 102, 20
    ref, qry = make_pairs(2000, q, c, seed=78, read_len=100, indel_rate=0.02)
    scoring = dict(match=7, mismatch=-11, gap_open=-19, gap_extend=-2)
    eng = _engine(q, c, match=7, mismatch=11, gap_read=19, gap_ref=19, gap_extend=2)
    for mode in (0, 1):
        _check(eng, mode, ref, qry, c, scoring)
    eng.close()


def test_affine_matches_reference_goldens():
    """Vectors captured from the reference's own EndToEndAffine (oracle/make_affine_goldens.py)."""
    g = np.load(GOLD, allow_pickle=False)
    ref, qry, c = g["ref"], g["qry"], int(g["c"])
    eng = _engine(qry.shape[1], c)
    for mode, mn in ((0, "local"), (1, "endfree")):
        valid = g[mn + "_valid"]
        sc = eng.BatchScore(mode, ref, qry)
        assert np.array_equal(sc[valid], g[mn + "_score"][valid])
        al = eng.BatchAlign(mode, ref, qry)
        want_cig = bytes(g[mn + "_cigars"]).split(b"\n")
        for i in np.nonzero(valid)[0]:
            a = al[i]
            assert a["cigar"] == want_cig[i], (i, a["cigar"], want_cig[i])
            assert (a["position_offset"], a["qstart"], a["qend"], a["nm"]) == tuple(
                int(g[mn + "_" + k][i]) for k in ("position_offset", "qstart", "qend", "nm"))
            assert np.float32(a["identity"]).view(np.uint32) == g[mn + "_identity"][i].view(np.uint32)
    eng.close()


def test_affine_needs_extend_penalty():
    import nextgenmap_amd as N
    from nextgenmap_amd import engine as E
    with pytest.raises(N.NgmHipError):
        N.Engine(152, 27, personality=E.PERSONALITY_AFFINE, gap_extend=0)
