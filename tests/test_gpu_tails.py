"""-m gpu: batch sizes that are not multiples of the kernels' block shapes (64 pairs per block, two blocks per wave in the
packed 16-bit kernels).  NextGenMap's ScoreBuffer / AlignmentBuffer submit whatever is in the buffer when a read batch ends
(src/ScoreBuffer.cpp:80-132), so BatchScore / BatchAlign see every n; found by the drop-in run of the real program
(tests/test_gpu_dropin.py)."""
import numpy as np
import pytest

import oracle_lib as O
from pairgen import make_pairs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("personality", ["linear", "affine"])
@pytest.mark.parametrize("q,c,rl", [(102, 20, 100), (152, 27, 150)])
def test_batch_score_every_tail_length(personality, q, c, rl):
    import nextgenmap_amd as N
    ref, qry = make_pairs(400, q, c, seed=4242 + q, read_len=rl)
    if personality == "linear":
        eng = N.Engine(q, c)
        want = {m: O.oracle_score(m, ref, qry, c, nthreads=8) for m in (0, 1)}
    else:
        eng = N.Engine(q, c, gap_read=33, gap_ref=33, gap_extend=3, personality=1)
        want = {m: O.oracle_affine(m, ref, qry, c, nthreads=8)[0] for m in (0, 1)}
    bad = []
    # ascending, then a shuffled order: a small batch after a large one sees what the large one left in the workspace
    rng = np.random.default_rng(99)
    order = list(range(1, 200)) + [255, 256, 257, 319, 320, 321, 383, 385, 400] + [int(x) for x in rng.integers(1, 401, 150)]
    for n in order:
        for mode in (0, 1):
            got = eng.BatchScore(mode, ref[:n], qry[:n])
            w = want[mode][:n]
            if not np.array_equal(np.asarray(got, np.float32), np.asarray(w, np.float32)):
                k = int(np.nonzero(np.asarray(got) != np.asarray(w))[0][0])
                bad.append((n, mode, k, float(got[k]), float(w[k])))
    eng.close()
    assert not bad, bad[:10]
