#!/usr/bin/env python3
"""bench.py -- mapped reads/s and SW Gcells/s of the MI355X mapping path.

One "step" = one pass of the hot path over one batch of R synthetic 150 bp reads whose bytes are already
resident in HBM: candidate search over the HBM-resident k-mer index of a synthetic GRCh38-sized genome
(seeded, with repeat families), device window gather, BatchScore over every candidate, top-1 selection +
MAPQ, BatchAlign (DP + traceback) of the winners, CIGAR/position on the host.  Shape: qry_max_len 152,
corridor 27, local mode, k 13 / kmer_skip 2 / bin_size 2, sensitivity 0.5 pinned.  Scoring personality:
--personality affine (default; `ngm --affine`, 10/15/33/3, the personality BASELINE.json's north star names and
the only one the reference program can run on this host, so cpu_baseline computes the SAME alignments and the
bench cross-checks its SAM records against ours) or linear (NGM's default OpenCL personality, 10/15/20/20).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--genome-mbp G] [--reads-per-step R]
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
          one rank per GPU; every rank holds the whole genome + index in its own HBM and maps its own shard of
          reads (weak scaling, no data-path collective); ONE RCCL all-reduce sums the mapping statistics.

Prints ONE JSON line (rank 0): `roofline` describes the dominant kernel (candidate search) from HIP events
recorded on the launch stream; `cpu_baseline` is the REAL reference program (NextGenMap's ngm-core built from
its sources by oracle/ngm_ref.mk, --affine because the default backend needs an OpenCL CPU device) run on this
host on a bounded sample of the same reads against the same genome (it loads the index cache files this
library writes), or, when that binary is absent, the oracle's C restatement of the score stage only.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

Q, C, READ_LEN, KMER = 152, 27, 150, 13
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.zeros(256, np.uint8)
COMP[list(b"ACGTN")] = list(b"TGCAN")


def make_genome(total_bp, seed):
    """24 contigs with GRCh38-like relative sizes, random sequence + repeat families (1 kb, 2 % diverged copies)
    + a few N runs.  Seed 20240601 as in SURVEY.md 8d."""
    rng = np.random.default_rng(seed)
    rel = np.array([248, 242, 198, 190, 181, 171, 159, 145, 138, 133, 135, 133, 114, 107, 102, 90, 83, 80, 59, 64, 47, 51, 156, 57], float)
    lens = np.maximum(20000, (rel / rel.sum() * total_bp).astype(np.int64))
    contigs = [ACGT[rng.integers(0, 4, int(n), dtype=np.uint8)] for n in lens]
    n_fam = max(4, int(total_bp // 2_000_000))
    for _ in range(n_fam):
        fam = ACGT[rng.integers(0, 4, 1000)]
        for _ in range(int(rng.integers(3, 30))):
            c = contigs[int(rng.integers(0, len(contigs)))]
            p = int(rng.integers(0, len(c) - 1000))
            cp = fam.copy()
            m = rng.random(1000) < 0.02
            cp[m] = ACGT[rng.integers(0, 4, int(m.sum()))]
            c[p:p + 1000] = cp
    for c in contigs:
        for _ in range(2):
            p = int(rng.integers(0, len(c) - 5000))
            c[p:p + int(rng.integers(50, 3000))] = ord("N")
    return contigs


def make_reads(contigs, n, seed, paired=False):
    """uniform positions, 50 % reverse strand, 1 % substitutions, one 1-3 bp indel in 15 % of the reads (0.1 % of the
    bases).  paired: rows 2i / 2i+1 are the two ends of a fragment, insert size ~ N(350, 35), FR orientation, half of
    the fragments from the reverse strand.  Returns ([n, Q] uint8 rows, truth contig, truth pos)."""
    rng = np.random.default_rng(seed)
    lens = np.array([len(c) for c in contigs], dtype=np.float64)
    rows = np.zeros((n, Q), np.uint8)
    pos = np.zeros(n, np.int64)
    if paired:
        nf = n // 2
        cf = rng.choice(len(contigs), size=nf, p=lens / lens.sum())
        ins = np.maximum(READ_LEN + 10, rng.normal(350, 35, nf).astype(np.int64))
        ci = np.repeat(cf, 2)
        flip = rng.random(nf) < 0.5
        for k in range(len(contigs)):
            sel = np.nonzero(cf == k)[0]
            if sel.size == 0:
                continue
            p = rng.integers(0, len(contigs[k]) - 600, sel.size)
            left = contigs[k][p[:, None] + np.arange(READ_LEN)[None, :]]
            pr = p + ins[sel] - READ_LEN
            right = COMP[contigs[k][pr[:, None] + np.arange(READ_LEN)[None, :]][:, ::-1]]
            f = flip[sel]
            rows[2 * sel, :READ_LEN] = np.where(f[:, None], right, left)
            rows[2 * sel + 1, :READ_LEN] = np.where(f[:, None], left, right)
            pos[2 * sel] = np.where(f, pr, p)
            pos[2 * sel + 1] = np.where(f, p, pr)
    else:
        ci = rng.choice(len(contigs), size=n, p=lens / lens.sum())
        for k in range(len(contigs)):
            sel = np.nonzero(ci == k)[0]
            if sel.size == 0:
                continue
            p = rng.integers(0, len(contigs[k]) - READ_LEN - 8, sel.size)
            pos[sel] = p
            rows[sel, :READ_LEN] = contigs[k][p[:, None] + np.arange(READ_LEN)[None, :]]
    sub = rng.random((n, READ_LEN)) < 0.01
    rows[:, :READ_LEN][sub] = ACGT[rng.integers(0, 4, int(sub.sum()))]
    for i in np.nonzero(rng.random(n) < 0.15)[0]:
        a = int(rng.integers(20, READ_LEN - 20))
        L = int(rng.integers(1, 4))
        r = rows[i, :READ_LEN].copy()
        if rng.random() < 0.5:  # insertion into the read
            rows[i, a + L:READ_LEN] = r[a:READ_LEN - L]
            rows[i, a:a + L] = ACGT[rng.integers(0, 4, L)]
        else:                   # deletion from the read
            rows[i, a:READ_LEN - L] = r[a + L:READ_LEN]
            rows[i, READ_LEN - L:READ_LEN] = ACGT[rng.integers(0, 4, L)]
    if not paired:
        rev = rng.random(n) < 0.5
        rows[rev, :READ_LEN] = COMP[rows[rev, :READ_LEN][:, ::-1]]
    return rows, ci, pos


def _sam_records(path, paired=False):
    recs = {}
    for line in open(path):
        if line.startswith("@"):
            continue
        f = line.split("\t")
        tags = {t[:2]: t[5:].strip() for t in f[11:]}
        idx = int(f[0][1:])
        if paired:  # r<pair> with the mate in the flag
            idx = 2 * idx + (1 if int(f[1]) & 0x80 else 0)
        recs[idx] = (int(f[1]), f[2], int(f[3]), int(f[4]), f[5], tags.get("AS"), tags.get("NM"))
    return recs


def cpu_baseline_reference(ref, rows, budget_reads, workdir, ours=None, paired=False):
    """NextGenMap itself (ngm-core --affine, host cores) on the first `budget_reads` reads vs the same genome.
    ours = (hits, cigar rows, contig names) of the GPU path in the affine personality: the reference's SAM records
    are then compared with them (flag, contig, position, MAPQ, CIGAR, AS, NM)."""
    import ref_files as RF
    cores = os.cpu_count() or 1
    fa = os.path.join(workdir, "bench_ref.fa")
    with open(fa, "w") as f:
        f.write(">stub\nACGT\n")  # with the caches present the program only checks that the file exists
    t = time.perf_counter()
    ref.write_ngm_cache(fa)
    t_cache = time.perf_counter() - t
    n = min(budget_reads, rows.shape[0]) & ~1
    fq, one = os.path.join(workdir, "sample.fq"), os.path.join(workdir, "one.fq")
    qual = b"I" * READ_LEN

    def name(i):
        return (b"@r%d/%d" % (i // 2, i % 2 + 1)) if paired else (b"@r%d" % i)
    with open(fq, "wb") as f:
        for i in range(n):
            f.write(name(i) + b"\n" + rows[i, :READ_LEN].tobytes() + b"\n+\n" + qual + b"\n")
    with open(one, "wb") as f:
        for i in range(2):
            f.write(name(i) + b"\n" + rows[i, :READ_LEN].tobytes() + b"\n+\n" + qual + b"\n")
    threads = min(cores, 64)

    def run(reads):
        cmd = [RF.NGM_CORE, "-r", fa, "-q", reads, "-o", os.path.join(workdir, "ref_out.sam"), "--affine", "-t", str(threads),
               "--no-progress", "-s", "0.5"] + (["-p"] if paired else [])
        t0 = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=workdir)
        dt = time.perf_counter() - t0
        if "Done" not in (r.stdout + r.stderr):
            raise RuntimeError("reference run failed: " + (r.stdout + r.stderr)[-400:])
        return dt

    t_load = run(one)   # index/genome load + start-up
    t_all = run(fq)
    t_map = max(t_all - t_load, 1e-3)
    parity = None
    if ours is not None:
        hits, cig, names = ours
        recs = _sam_records(os.path.join(workdir, "ref_out.sam"), paired)
        same = same_place = cmp = 0
        examples = []
        for i, (flag, rname, pos, mapq, cigar, a_s, nm) in recs.items():
            h = hits[i]
            if (flag & 4) or not h["mapped"]:
                continue  # the writer's identity / residue filter is not part of the timed path
            cmp += 1
            mine = (16 if h["reverse"] else 0, names[h["contig"]], int(h["pos"]) + 1, int(h["mapq"]),
                    bytes(cig[i]).split(b"\0", 1)[0].decode(), str(int(h["score"])), str(int(h["nm"])))
            same += mine == (flag & 16, rname, pos, mapq, cigar, a_s, nm)
            if len(examples) < 3 and mine != (flag & 16, rname, pos, mapq, cigar, a_s, nm):
                examples.append({"read": i, "ours": mine, "reference": (flag & 16, rname, pos, mapq, cigar, a_s, nm)})
            same_place += mine[:3] == (flag & 16, rname, pos)
        parity = {"reads_compared": cmp, "identical_records": same, "same_position": same_place, "first_differences": examples,
                  "note": "all SAM fields compared for reads both sides report as mapped; equal scores are resolved in the reference's own candidate order (cs_order_kernel)"}
    return {"parity_vs_reference_sam": parity, "value": n / t_map, "unit": "reads/s", "cores": threads, "kind": "reference",
            "sample": "NextGenMap 0.5.5 ngm-core --affine " + ("-p " if paired else "") + "-t %d on the first %d reads of the step vs the same genome (index "
                      "loaded from cache files written by this library): %.1fs total minus %.1fs index load/start-up measured "
                      "with a 1-read run" % (threads, n, t_all, t_load),
            "index_cache_write_s": t_cache}


def cpu_baseline_port(rows_qry, budget_s=8.0):
    import oracle_lib as O
    cores = os.cpu_count() or 1
    rng = np.random.default_rng(0)
    wins = ACGT[rng.integers(0, 4, (len(rows_qry), Q + C), dtype=np.uint8)]
    n0 = min(len(rows_qry), max(2048, 32 * cores))
    O.oracle_score(0, wins[:256], rows_qry[:256], C, nthreads=cores)
    t = time.perf_counter()
    O.oracle_score(0, wins[:n0], rows_qry[:n0], C, nthreads=cores)
    dt = max(time.perf_counter() - t, 1e-4)
    reps = int(max(1, min(1024, budget_s / dt)))
    idx = np.arange(n0 * reps) % len(rows_qry)
    t = time.perf_counter()
    O.oracle_score(0, wins[idx], rows_qry[idx], C, nthreads=cores)
    ts = time.perf_counter() - t
    return {"value": len(idx) / ts, "unit": "scored pairs/s", "cores": cores, "kind": "port",
            "sample": "%d pairs, oracle C restatement of BatchScore only, OpenMP %d threads (no candidate search)" % (len(idx), cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--genome-mbp", type=float, default=3100.0, help="synthetic genome size (GRCh38 = 3100)")
    ap.add_argument("--reads-per-step", type=int, default=1 << 20, help="reads per GPU per step")
    ap.add_argument("--cpu-sample-reads", type=int, default=200000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--personality", choices=["affine", "linear"], default="affine")
    ap.add_argument("--workers", type=int, default=2, help="mapper instances (streams + host threads) per GPU")
    ap.add_argument("--read-len", type=int, default=150, help="read length (150: BASELINE config #2/#3; 250: config #5's shape)")
    ap.add_argument("--corridor", type=int, default=0, help="band width; 0: NextGenMap's 5 + 0.15 * read length")
    ap.add_argument("--layout", choices=["pe", "se"], default="pe", help="paired-end (BASELINE.json config #2) or single-end reads")
    args = ap.parse_args()
    global Q, C, READ_LEN
    READ_LEN = args.read_len
    Q = (READ_LEN | 1) + 1                                   # ReadProvider.cpp:288
    C = args.corridor if args.corridor > 0 else int(5 + 0.15 * READ_LEN)  # ReadProvider.cpp:304

    import torch
    import torch.distributed as dist
    from nextgenmap_amd.pipeline import HIT_DTYPE, Mapper, Reference

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    R = args.reads_per_step
    t0 = time.perf_counter()
    contigs = make_genome(int(args.genome_mbp * 1e6), seed=20240601)
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    ref = Reference.from_contigs(contigs, device=local_rank, kmer=KMER, kmer_skip=2, bin_size=2)
    t_index = time.perf_counter() - t0
    paired = args.layout == "pe"
    rows, truth_c, truth_p = make_reads(contigs, R, seed=20240602 + 2 + 1000 * rank, paired=paired)  # config #2's seed, one shard per rank
    d_rows = torch.from_numpy(rows).to(dev)
    affine = args.personality == "affine"
    band = C + 1 if affine else C  # SeqAn's band has diagonals 0..corridor
    # W mapper instances (own stream + workspace each, like NextGenMap's CS threads with their own IAlignment) work on
    # contiguous slices of the step's reads from W host threads: the host stages of one slice (pair selection, CIGAR,
    # downloads) overlap the kernels of the others
    W = max(1, min(args.workers, R // 2048))
    bounds = [(R * w // W) & ~1 for w in range(W)] + [R]
    out = (np.zeros(R, HIT_DTYPE), np.zeros((R, 4 * Q), np.uint8), np.zeros((R, 4 * Q), np.uint8))
    mps, views = [], []
    for w in range(W):
        kw = dict(gap_read=33, gap_ref=33, gap_extend=3, personality=1) if affine else {}
        mps.append(Mapper(ref, Q, C, sensitivity=0.5, **kw))
        lo, hi = bounds[w], bounds[w + 1]
        views.append((rows[lo:hi], d_rows[lo:hi], tuple(o[lo:hi] for o in out)))

    def worker(w, steps, acc):
        rw, dw, ow = views[w]
        k = np.zeros(8)
        for _ in range(steps):
            if paired:
                mps[w].map_pe_raw(rw, dw, ow)
            else:
                mps[w].map_se_raw(rw, dw, ow)
            k += np.array(mps[w].last_kernel_ms())
        acc[w] = k

    def run(steps):
        import threading
        acc = [None] * W
        ts = [threading.Thread(target=worker, args=(w, steps, acc)) for w in range(W)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        return np.sum(acc, axis=0)

    run(args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kms = run(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kms /= max(1, args.steps)  # GPU time per step (R reads), summed over the W streams' launches
    hits = out[0]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # one extra, untimed pass of instance 0 alone: kernel durations without the other streams' kernels sharing the GPU
    worker(0, 1, iso := [None] * W)
    iso_ms = iso[0]
    iso_ctr = mps[0].cs_counters()
    ctr = np.sum([m_.cs_counters() for m_ in mps], axis=0)
    kmers, hits_voted, n_cand = int(ctr[0]), int(ctr[1]), int(ctr[2])
    mapped = hits["mapped"] == 1
    correct = mapped & (hits["contig"] == truth_c) & (np.abs(hits["pos"].astype(np.int64) - truth_p) <= C // 2 + 4)
    stats = torch.tensor([R, int(mapped.sum()), int((~mapped).sum()), int(mapped.sum()), int(correct.sum()), int((hits["mapq"] > 0).sum()),
                          int(n_cand), int(hits_voted)], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)  # the ONE collective of the path (RCCL over xGMI)
    stats = [int(x) for x in stats.tolist()]

    if rank == 0:
        value = R * world * args.steps / elapsed
        score_cells = n_cand * READ_LEN * band
        align_cells = int(mapped.sum()) * READ_LEN * band
        b_cs = 20 * kmers + 4 * hits_voted + 16 * n_cand
        achieved = b_cs / (kms[0] * 1e-3) / 1e9
        # HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/*_pmc_traffic.json, keyed
        # by kernel name and grid size = 64 threads per read; collected with this default workload)
        traffic = None
        if int(args.genome_mbp) == 3100 and READ_LEN == 150:
            for fn in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
                if fn.endswith("_pmc_traffic.json"):
                    try:
                        tj = json.load(open(os.path.join(ROOT, "profiles", fn)))
                    except Exception:
                        continue
                    for k, v in tj.items():
                        if k.startswith("ngm::cs_fast_kernel") and k.endswith("|grid=%d" % ((bounds[1] - bounds[0]) * 64)):
                            traffic = v
                    if traffic is not None:
                        break
        line = {
            "metric": "mapped reads/sec + SW Gcells/sec, 150bp vs GRCh38, at 1/2/4/8 MI355X",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": ("%d x %dbp %s synthetic reads per GPU per step vs a synthetic %.0f Mbp genome (24 contigs, repeat "
                                    "families, N runs; GRCh38 itself is not available offline): candidate search (k=13, skip 2, s=0.5) + "
                                    "score + %s/MAPQ + align with traceback + CIGAR; reads resident in HBM")
                       % (R, READ_LEN, "PE (insert ~N(350,35), FR)" if paired else "SE", args.genome_mbp, "pair selection (top1PE)" if paired else "top-1"),
                       "qry_max_len": Q, "corridor": C, "scoring": "affine (SeqAn banded Gotoh) 10/15/33/3 local" if affine else "linear 10/15/20/20 local", "reads_per_step_per_gpu": R, "mapper_instances_per_gpu": W,
                       "parallelism": "reads sharded x%d, genome+index replicated per GPU" % world},
            "sw_gcells_per_s": {"score_kernel": score_cells / (kms[2] * 1e-3) / 1e9 if kms[2] > 0 else None,
                                "align_kernel": align_cells / (kms[5] * 1e-3) / 1e9 if kms[5] > 0 else None},
            "kernel_ms": {"candidate_search": kms[0], "gather_score": kms[1], "sw_score": kms[2], "select": kms[3], "gather_align": kms[4],
                          "sw_align": kms[5], "traceback": kms[6], "all_kernels": float(kms[:7].sum()),
                          "candidate_search_stage_incl_host_sync": kms[7]},
            "per_read": {"candidates": n_cand / R, "index_hits": hits_voted / R, "kmers": kmers / R},
            "accuracy": {"mapped": stats[1] / stats[0], "within_band_of_truth": stats[4] / stats[0], "mapq_gt0": stats[5] / stats[0]},
            "setup_s": {"genome_generation": t_gen, "encode+index_build": t_index, "index_entries": ref.index_entries},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "cs_fast_kernel (candidate search)",
                         "isolated": {"achieved": (20 * iso_ctr[0] + 4 * iso_ctr[1] + 16 * iso_ctr[2]) / (iso_ms[0] * 1e-3) / 1e9, "ms": float(iso_ms[0]),
                                      "reads": int(bounds[1] - bounds[0]),
                                      "note": "same kernel, one launch of one mapper instance with nothing else on the GPU (untimed extra pass)"}, "bytes_per_launch": b_cs / W, "launches_per_step": W,
                         "note": "algorithmic bytes = 20 B/k-mer + 4 B/index hit + 16 B/candidate (SURVEY.md 8d); dependent random "
                                 "4-16 B gathers; the kernel is bound by LDS atomics and instruction issue rather than HBM (DESIGN.md 4); the SW kernels are "
                                 "VALU-bound, see sw_gcells_per_s"},
            # SURVEY.md 8d's whole-path figure: (pairs * B_score + alignments * B_align + B_cs) per second of wall time
            "path_algorithmic_gbs": (n_cand * (Q + Q + C + 4) + int(mapped.sum()) * (Q + Q + C + 8 + 4 * (2 * Q + C + 1)) + b_cs) * world
                                    / (elapsed / args.steps) / 1e9,
            "stats_allreduce": {"reads": stats[0], "mapped": stats[1], "unmapped": stats[2], "candidates": stats[6]},
        }
        if world > 1:
            line["cpu_baseline"] = None  # the host baseline is timed on rank 0 of a 1-GPU run only
        elif not args.no_cpu_baseline:
            try:
                import ref_files as RF
                if not RF.have_reference_binary():
                    raise RuntimeError("oracle/_ref/ngm/ngm-core not built")
                with tempfile.TemporaryDirectory() as wd:
                    ours = (hits, out[1], [c[0] for c in ref.contigs]) if affine else None
                    line["cpu_baseline"] = cpu_baseline_reference(ref, rows, args.cpu_sample_reads, wd, ours, paired)
            except Exception as e:  # the port of the score stage only
                line["cpu_baseline"] = cpu_baseline_port(rows[:8192])
                line["cpu_baseline"]["note"] = "reference program unavailable: %s" % str(e)[:200]
        print(json.dumps(line))
    for m_ in mps:
        m_.close()
    ref.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
