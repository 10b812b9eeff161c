#!/usr/bin/env python3
"""Generate golden vectors by RUNNING THE REFERENCE'S OWN KERNELS on the MI355X.

Runs on a GPU box (gpurun): loads oracle/_ref/*.co -- NextGenMap's OpenCL kernels compiled
unmodified for gfx950 by oracle/build_ref.sh -- through oracle/libngm_ref_runner.so, feeds them the
seeded pairs of tests/pairgen.py, and writes inputs + raw kernel outputs as compressed .npz files
(copied afterwards to tests/golden/ and committed).  It also diffs every output against the C
restatement (oracle/libngm_oracle.so) and prints the mismatch counts: this is how the oracle is
pinned.  Reference sources never leave /root/reference; only these data files are committed.

usage: python oracle/make_goldens.py [outdir]      (default gpurun_out/golden)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from pairgen import make_alt_pairs, make_pairs  # noqa: E402

# (variant, q, c, n, read_len, seed)
CASES = [
    (0, 32, 8, 3000, 28, 11),
    (0, 102, 20, 2048, 100, 12),
    (0, 152, 27, 2048, 150, 13),
    (0, 252, 42, 768, 250, 14),
    (0, 252, 80, 512, 250, 15),
    (1, 32, 8, 1000, 28, 16),
    (1, 102, 20, 1000, 100, 17),
]


# the -D__ALT_SCORING__ builds (`--bs-mapping`, `--slam-seq 2`): (variant, q, c, n, read_len, seed, tag, scoring)
ALT_CASES = [
    (0, 102, 20, 2048, 100, 21, "bs", O.BS_SCORING),
    (0, 152, 27, 2048, 150, 22, "bs", O.BS_SCORING),
    (1, 102, 20, 1000, 100, 23, "bs", O.BS_SCORING),
    (0, 102, 20, 2048, 100, 24, "slam", O.SLAM_SCORING),
]


def compare(name, a, b):
    bad = int(np.count_nonzero(a != b))
    print("  %-28s mismatches: %d / %d" % (name, bad, a.size))
    return bad


def main():
    outdir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "golden")
    os.makedirs(outdir, exist_ok=True)
    total_bad = 0
    for case in CASES + ALT_CASES:
        variant, q, c, n, read_len, seed = case[:6]
        tag, scoring = (case[6], case[7]) if len(case) > 6 else ("", None)
        vname = "gpu" if variant == 0 else "cpu"
        print("case %s q=%d c=%d n=%d %s" % (vname, q, c, n, tag))
        dirs = None
        if scoring:
            ref, qry, dirs = make_alt_pairs(n, q, c, seed=seed, read_len=read_len, alt=scoring["alt"])
        else:
            ref, qry = make_pairs(n, q, c, seed=seed, read_len=read_len)
        if variant == 1:
            # the float4 build decides "skip" from the first pair of each group of four
            # (oclSwScore.cl:124); keep empty reads out of the goldens for that build
            empty = qry[:, 0] == 0
            qry[empty, 0] = ord('A')
            qry[:4, :] = 0  # ... except one whole group of four empty reads
        al = 2 * q + c + 1
        out = dict(ref=ref, qry=qry, q=q, c=c, variant=variant, seed=seed)
        if scoring:
            out.update(dirs=dirs, alt=scoring["alt"], scoring=np.array([scoring[k] for k in ("match", "mismatch", "gap_read", "gap_ref", "alt", "match_alt", "mismatch_alt")], np.int32))
        for mode in (0, 1):
            mname = "local" if mode == 0 else "endfree"
            sc, ms_s = O.ref_score(variant, mode, ref, qry, c, scoring, dirs)
            # end-to-end align of the 256-work-item __GPU__ build is racy in the reference
            # (oclEndFreeScore.cl:229); take that golden from the one-work-item build of the same source
            av = 2 if (variant == 0 and mode == 1) else variant
            res, rle, ms_a = O.ref_align(av, mode, ref, qry, c, scoring, dirs)
            if variant == 0 and mode == 1:
                res256, _, _ = O.ref_align(0, mode, ref, qry, c, scoring, dirs)
                print("  [endfree] rows where the 256-wide build differs from the 1-wide build of the same kernel: %d"
                      % int(np.count_nonzero((res256 != res).any(axis=1))))
                sc1, _ = O.ref_score(2, 0, ref, qry, c, scoring, dirs)
                sc0, _ = O.ref_score(0, 0, ref, qry, c, scoring, dirs)
                print("  [local] score, 1-wide vs 256-wide build mismatches: %d" % int(np.count_nonzero(sc1 != sc0)))
            print("  [%s] reference kernels: score %.3f ms, align+backtrack %.3f ms" % (mname, ms_s, ms_a))
            o_sc = O.oracle_score(mode, ref, qry, c, scoring, variant=variant, dirs=dirs)
            o_res, o_rle, o_valid, o_best = O.oracle_trace(mode, ref, qry, c, scoring, variant=variant, dirs=dirs)
            total_bad += compare(mname + " score", sc, o_sc)
            # rows where the reference skipped backtracking leave res[3]/rle untouched (we memset 0)
            total_bad += compare(mname + " result[0:3]", res[:, :3], o_res[:, :3])
            total_bad += compare(mname + " alignment_offset (valid)", res[o_valid, 3], o_res[o_valid, 3])
            # compare only the written region [offset, al-1] of each rle row
            bad = 0
            for i in np.nonzero(o_valid)[0]:
                off = int(o_res[i, 3])
                if off != int(res[i, 3]) or not np.array_equal(rle[i, off:al], o_rle[i, off:al]):
                    bad += 1
            print("  %-28s mismatches: %d / %d" % (mname + " rle rows (valid)", bad, int(o_valid.sum())))
            total_bad += bad
            out[mname + "_score"] = sc
            out[mname + "_res"] = res
            out[mname + "_rle"] = rle[:, :al]  # second half of the 2*al buffer is never written
            out[mname + "_valid"] = o_valid
        np.savez_compressed(os.path.join(outdir, "ngm_ocl_%s_q%d_c%d%s.npz" % (vname, q, c, "_" + tag if tag else "")), **out)
    print("TOTAL MISMATCHES oracle vs reference kernels: %d" % total_bad)
    return 0 if total_bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
