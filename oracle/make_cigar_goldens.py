#!/usr/bin/env python3
"""Golden CIGAR / MD / NM / Identity / QStart / QEnd of NextGenMap's DEFAULT (linear-gap) personality, produced by
the REFERENCE'S OWN SWOclCigar::computeCigarMD (lib/mason/opencl/SWOclCigar.cpp:430-615).

Runs in the build container (needs /root/reference): oracle/_ref/ngm/ngm_linear_cigar_ref is the reference's
SWOclCigar.cpp compiled unmodified by oracle/ngm_ref.mk behind oracle/linear_cigar_ref_main.cpp.  Its input is what the
reference's own kernels produced on the MI355X -- the (result, RLE) rows of tests/golden/ngm_ocl_*.npz -- in the three
clipping modes (soft, --hard-clip, --silent-clip).  Output: tests/golden/ngm_ocl_*_cigar.npz (data only), which pins
oracle/ngm_oracle.c's restatement (tests/test_oracle_golden.py) and the product (tests/test_gpu_parity.py).

usage: python oracle/make_cigar_goldens.py [log]
"""
import glob
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
DRIVER = os.path.join(ROOT, "oracle", "_ref", "ngm", "ngm_linear_cigar_ref")
CLIPS = ("soft", "hard", "silent")


def run_reference(ref, qry, c, res, rle, rows, clip, alt=0, dirs=None):
    """-> dict of arrays over `rows` as the reference's computeCigarMD returns them"""
    n, q = len(rows), qry.shape[1]
    al = 2 * q + c + 1
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.txt")
        with open(fin, "wb") as f:
            f.write(np.array([n, q, c], np.int32).tobytes())
            for i in rows:
                f.write(ref[i, :q + c].tobytes()); f.write(qry[i].tobytes())
                f.write(res[i].astype(np.int16).tobytes()); f.write(rle[i, :al].astype(np.int16).tobytes())
                if alt:
                    f.write(bytes([int(dirs[i])]))
        subprocess.check_call([DRIVER, fin, fout, str(clip)] + ([str(alt)] if alt else []), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lines = open(fout, "rb").read().split(b"\n")[:n]
    out = dict(ok=np.zeros(n, np.int8), cigar=[], md=[], nm=np.zeros(n, np.int32), identity=np.zeros(n, np.float32),
               qstart=np.zeros(n, np.int32), qend=np.zeros(n, np.int32), position_offset=np.zeros(n, np.int32),
               score_token=np.zeros(n, np.float32))
    for j, ln in enumerate(lines):
        t = ln.split(b"\t")
        assert int(t[0]) == j and len(t) == 10, ln
        out["ok"][j] = int(t[1]); out["cigar"].append(t[2]); out["md"].append(t[3]); out["nm"][j] = int(t[4])
        out["identity"][j] = np.float32(float(t[5])); out["qstart"][j] = int(t[6]); out["qend"][j] = int(t[7])
        out["position_offset"][j] = int(t[8]); out["score_token"][j] = np.float32(float(t[9]))
    out["cigar"] = np.array(out["cigar"], dtype="S"); out["md"] = np.array(out["md"], dtype="S")
    return out


def main():
    import oracle_lib as O
    log = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
    total_bad = 0
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "ngm_ocl_*.npz"))):
        if path.endswith("_cigar.npz"):
            continue
        g = np.load(path)
        ref, qry, c, variant = g["ref"], g["qry"], int(g["c"]), int(g["variant"])
        alt = int(g["alt"]) if "alt" in g.files else 0
        dirs = g["dirs"] if alt else None
        scoring = dict(zip(("match", "mismatch", "gap_read", "gap_ref", "alt", "match_alt", "mismatch_alt"), (int(x) for x in g["scoring"]))) if alt else None
        out = {}
        for mode, mn in ((0, "local"), (1, "endfree")):
            rows = np.nonzero(g[mn + "_valid"])[0]
            out[mn + "_rows"] = rows.astype(np.int32)
            for clip, cn in enumerate(CLIPS):
                r = run_reference(ref, qry, c, g[mn + "_res"], g[mn + "_rle"], rows, clip, alt, dirs)
                for k, v in r.items():
                    out["%s_%s_%s" % (mn, cn, k)] = v
                # the C restatement on the same pairs (it recomputes the DP: its RLE equals the golden RLE, test_oracle_golden.py)
                res, cig, md = O.oracle_align(mode, ref, qry, c, scoring, variant=variant, hard_clip=int(clip == 1), silent_clip=int(clip == 2), nthreads=8, dirs=dirs)
                bad = 0
                for j, i in enumerate(rows):
                    have = (bool(res["ok"][i]), cig[i], md[i], int(res["nm"][i]), np.float32(res["identity"][i]).tobytes(), int(res["qstart"][i]),
                            int(res["qend"][i]), int(res["position_offset"][i]), float(res["score_token"][i]))
                    want = (bool(r["ok"][j]), bytes(r["cigar"][j]), bytes(r["md"][j]), int(r["nm"][j]), np.float32(r["identity"][j]).tobytes(),
                            int(r["qstart"][j]), int(r["qend"][j]), int(r["position_offset"][j]), float(r["score_token"][j]))
                    if have != want:
                        if bad < 3:
                            print("   MISMATCH row %d: oracle %r reference %r" % (i, have, want), file=log)
                        bad += 1
                print("%-28s %-8s %-6s computeCigarMD rows: %5d  oracle-vs-reference mismatches: %d" %
                      (os.path.basename(path), mn, cn, len(rows), bad), file=log)
                total_bad += bad
        np.savez_compressed(path.replace(".npz", "_cigar.npz"), **out)
    print("TOTAL MISMATCHES oracle vs reference computeCigarMD: %d" % total_bad, file=log)
    return 0 if total_bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
