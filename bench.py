#!/usr/bin/env python3
"""bench.py -- throughput of the score/align hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input that is already resident in
HBM: BatchScore over R*cpr candidate (read, window) pairs, then BatchAlign (DP + traceback) over the R
winning pairs -- the work NextGenMap's ScoreBuffer / AlignmentBuffer stages do for R reads
 at the 150 bp shape (qry_max_len 152, corridor 27, scoring 10/15/20/20).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
          one rank per GPU, reads sharded across ranks (weak scaling: R reads per GPU per step), no
          data-path collective; the mapping-stats vector is summed with one RCCL all-reduce at the end.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the score DP) from HIP events
recorded on the launch stream inside this script; `cpu_baseline` is the oracle's C restatement of the
same arithmetic timed on this host's cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

Q, C, READ_LEN = 152, 27, 150
CPR = 4  # candidate windows scored per read
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
B_SCORE = Q + (Q + C) + 4  # algorithmic bytes per scored pair   (SURVEY.md 8d): 335
B_ALIGN = Q + (Q + C) + 8 + 4 * (2 * Q + C + 1)  # per aligned pair: 1667


def make_pool(n_pool, seed):
    from pairgen import make_pairs
    return make_pairs(n_pool, Q, C, seed=seed, read_len=READ_LEN, mix=(0.55, 0.40, 0.05))


def cpu_baseline(pool_ref, pool_qry, budget_s=12.0):
    """Oracle (kind 'port') on all host cores, bounded sample of the same workload."""
    import oracle_lib as O
    cores = os.cpu_count() or 1
    n0 = min(len(pool_qry), max(2048, 32 * cores))
    O.oracle_score(0, pool_ref[:256], pool_qry[:256], C, nthreads=cores)  # spin the thread pool up
    t = time.perf_counter()
    O.oracle_score(0, pool_ref[:n0], pool_qry[:n0], C, nthreads=cores)
    O.oracle_align(0, pool_ref[:n0 // CPR], pool_qry[:n0 // CPR], C, nthreads=cores)
    dt = max(time.perf_counter() - t, 1e-4)
    reps = int(max(1, min(4096, budget_s / dt)))
    n = n0 * reps
    idx = np.arange(n) % len(pool_qry)
    ref, qry = pool_ref[idx], pool_qry[idx]
    t = time.perf_counter()
    O.oracle_score(0, ref, qry, C, nthreads=cores)
    ts = time.perf_counter() - t
    t = time.perf_counter()
    O.oracle_align(0, ref[:n // CPR], qry[:n // CPR], C, nthreads=cores)
    ta = time.perf_counter() - t
    reads = n // CPR
    return {"value": reads / (ts + ta), "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": "%d reads: %d scored pairs (%.2fs) + %d aligned pairs (%.2fs), oracle C restatement, OpenMP %d threads"
                      % (reads, n, ts, reads, ta, cores),
            "sw_gcells_per_s": n * READ_LEN * C / ts / 1e9}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads-per-step", type=int, default=1 << 20, help="reads per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import nextgenmap_amd as N

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
    NS = R * CPR
    pool_ref, pool_qry = make_pool(16384, seed=20240602 + rank)
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    idx = torch.randint(0, pool_ref.shape[0], (NS,), generator=g).to(dev)
    p_ref = torch.from_numpy(pool_ref).to(dev)
    p_qry = torch.from_numpy(pool_qry).to(dev)
    d_ref = p_ref.index_select(0, idx).contiguous()
    d_qry = p_qry.index_select(0, idx).contiguous()
    # the "winning" candidate of read r is pair r*CPR: gather once, outside the timed region (NGM's
    # align stage re-gathers windows on the host; that gather is not part of this bench yet)
    a_ref = d_ref[::CPR].contiguous()
    a_qry = d_qry[::CPR].contiguous()
    d_scores = torch.empty(NS, dtype=torch.float32, device=dev)
    eng = N.Engine(Q, C, device=local_rank, max_batch=NS)
    rs = eng.align_run_stride()
    d_rec = torch.empty((R, 8), dtype=torch.int32, device=dev)
    d_runs = torch.empty((R, rs), dtype=torch.int16, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        eng.score_device(N.MODE_LOCAL, NS, d_ref, d_qry, d_scores, stream)
        eng.align_device(N.MODE_LOCAL, R, a_ref, a_qry, d_rec, d_runs, rs, stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()

    # kernel durations: HIP events on the launch stream (the engine brackets its launches)
    eng.set_profiling(True)
    k_pack = k_score = k_apack = k_align = k_tb = 0.0
    nprof = 5
    for _ in range(nprof):
        eng.score_device(N.MODE_LOCAL, NS, d_ref, d_qry, d_scores, stream)
        torch.cuda.synchronize()
        ms = eng.last_kernel_ms()
        k_pack += ms[0]; k_score += ms[1]
        eng.align_device(N.MODE_LOCAL, R, a_ref, a_qry, d_rec, d_runs, rs, stream)
        torch.cuda.synchronize()
        ms = eng.last_kernel_ms()
        k_apack += ms[0]; k_align += ms[1]; k_tb += ms[2]
    eng.set_profiling(False)
    k_pack, k_score, k_apack, k_align, k_tb = (x / nprof for x in (k_pack, k_score, k_apack, k_align, k_tb))

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # mapping statistics of the last step, summed over ranks with ONE collective
    valid = d_rec[:, 0].to(torch.int64)
    stats = torch.stack([torch.tensor(R, device=dev, dtype=torch.int64), valid.sum(), (1 - valid).sum(),
                         d_scores.to(torch.int64).sum(), torch.tensor(NS, device=dev, dtype=torch.int64),
                         d_rec[:, 5].to(torch.int64).sum(), torch.zeros((), device=dev, dtype=torch.int64),
                         torch.zeros((), device=dev, dtype=torch.int64)])
    if world > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)  # the ONE collective of the path (RCCL over xGMI)
    stats = [int(x) for x in stats.tolist()]

    if rank == 0:
        reads_total = R * world * args.steps
        value = reads_total / elapsed
        score_cells = NS * READ_LEN * C
        align_cells = R * READ_LEN * C
        achieved = NS * B_SCORE / (k_score * 1e-3) / 1e9
        # HBM bytes per launch of the same kernel at the same grid, from the PMC passes committed under
        # profiles/ (FETCH_SIZE / WRITE_SIZE, separate runs, gfx950 read correction applied there)
        traffic = None
        for fn in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
            if fn.endswith("_pmc_traffic.json"):
                try:
                    tr = json.load(open(os.path.join(ROOT, "profiles", fn)))
                    traffic = tr.get("ngm::sw_score_kernel<%d, false>|grid=%d" % (C, ((NS + 255) // 256) * 256))
                except Exception:
                    traffic = None
                if traffic is not None:
                    break
        line = {
            "metric": "mapped reads/sec + SW Gcells/sec, 150bp vs GRCh38, at 1/2/4/8 MI355X",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "score+align stages at the 150bp shape (qry_max_len 152, corridor 27, linear gaps "
                                   "10/15/20/20, local mode): per GPU per step BatchScore over %d candidate pairs (%d reads x %d "
                                   "candidates) + BatchAlign with traceback over %d pairs; candidate search over a GRCh38 index is "
                                   "not in this bench yet" % (NS, R, CPR, R),
                       "reads_per_step_per_gpu": R, "candidates_per_read": CPR, "parallelism": "reads sharded x%d" % world},
            "sw_gcells_per_s": {"score_kernel": score_cells / (k_score * 1e-3) / 1e9,
                                "align_kernel": align_cells / (k_align * 1e-3) / 1e9,
                                "whole_step": (score_cells + align_cells) * world * args.steps / elapsed / 1e9},
            "kernel_ms": {"pack(score batch)": k_pack, "sw_score": k_score, "pack(align batch)": k_apack,
                          "sw_align": k_align, "traceback": k_tb},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "sw_score_kernel<27,local>", "bytes_per_launch": NS * B_SCORE,
                         "note": "integer-VALU bound kernel: %.0f Gcells/s; HBM fraction reported per contract"
                                 % (score_cells / (k_score * 1e-3) / 1e9)},
            "stats_allreduce": {"reads": stats[0], "aligned": stats[1], "no_alignment": stats[2], "score_sum": stats[3]},
        }
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(pool_ref, pool_qry)
        print(json.dumps(line))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
