"""Debug helper (not a test): the reads on which the real program + plugin (linear personality) and ngm-hip disagree in MAPQ.
Lists the candidates of those reads and their linear scores through three routes: the host-pointer BatchScore with windows
from ngm_ref_decode (what the plugin sees), the C oracle on the same windows, and what the device pipeline reported."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import simulate as S
import oracle_lib as O
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import nextgenmap_amd as N
from nextgenmap_amd.pipeline import Mapper, Reference

contigs = S.make_genome([400000, 300001], seed=901, repeat_families=10, repeat_len=500, copies=6)
reads = S.make_reads(contigs, 4000, 100, seed=903, sub_rate=0.02, indel_rate=0.003)
names = [r[0] for r in reads]
want = [n for n in names if n.startswith("r744_") or n.startswith("r1768_")]
q, c = 102, 20
ref = Reference.from_contigs(contigs, device=0)
rows = Mapper.reads_to_rows([r[1] for r in reads], q)
m = Mapper(ref, q, c, sensitivity=0.5)   # NB: the CLI estimates the sensitivity; close enough for the candidate list
offs, mx, loc, strand, votes = m.candidate_search(rows)
hits, cig, md = m.map_se(rows)
eng = N.Engine(q, c)
comp = np.zeros(256, np.uint8); comp[list(b"ACGTN")] = list(b"TGCAN")
for nm in want:
    i = names.index(nm)
    print("read", nm, "pipeline: mapq", hits[i]["mapq"], "score", hits[i]["score"], "n_cand", hits[i]["n_candidates"], "n_best", hits[i]["n_best"])
    L = int(np.count_nonzero(rows[i]))
    for k in range(offs[i], offs[i + 1]):
        win = np.frombuffer(ref.decode(int(loc[k]) - c // 2, ((q + c) | 1) + 1)[1], np.uint8)[:q + c].copy()
        qry = rows[i].copy()
        if strand[k]:
            qry[:L] = comp[qry[:L][::-1]]
        s_host = eng.BatchScore(0, win[None, :], qry[None, :])[0]
        s_or = O.oracle_score(0, win[None, :], qry[None, :], c)[0]
        print("   cand loc", int(loc[k]), "strand", int(strand[k]), "votes", float(votes[k]), "host BatchScore", float(s_host), "oracle", float(s_or), bytes(win[:40]))
