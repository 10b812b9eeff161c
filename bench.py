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

          Rank 0 builds the genome + index once and writes NextGenMap's cache files; the other ranks load those
          (nextgenmap_amd.sharding: shard_range / reduce_stats are the same two functions `ngm-hip` semantics follow).

Prints ONE JSON line (rank 0): `roofline` describes the dominant kernel (candidate search) from HIP events
recorded on the launch stream; `cpu_baseline` is the REAL reference program (NextGenMap's ngm-core built from
its sources by oracle/ngm_ref.mk, --affine because the default backend needs an OpenCL CPU device) run on this
host on a bounded sample of the same reads against the same genome (it loads the index cache files this
library writes), or, when that binary is absent, the oracle's C restatement of the score stage only.
`end_to_end` (N = 1, unless --no-end-to-end) is the DROP-IN measured: the `ngm-hip` program from the first input byte to
the closed SAM file -- FASTQ parsing, H2D copies, mapping, SAM text -- on --e2e-reads reads written as two plain FASTQ
files, index loaded from the cache files, next to `ngm-core -t <cores>` on a slice of the same files.

  --stub-mapper: no GPU, no library -- the mapping call is replaced by a deterministic fake so that the N > 1 control path
  (sharding, cache hand-over between ranks, stats all-reduce, the JSON line) runs under gloo on CPU (tests/test_sharding_gloo.py).
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
# roofline.traffic: the committed PMC pass of this round's default workload, and the kernel instance that workload launches (mapper.cpp cs_canon_fn)
PMC_TRAFFIC_FILE = "r06_mapping_pe_affine_pmc_traffic.json"
HEAVY_CS_TRAFFIC_SEARCHES = 4   # mapper instances of the PMC passes (profiles/tools/heavy_leg_only.py): their step of 1 048 576 reads is this many searches
HEAVY_CS_TRAFFIC_FILE = "r06_heavy_tail_%s_cs_traffic.json"   # per sub-leg (uniform / repeats): profiles/summarize_rocprof.py, from PMC passes of profiles/tools/heavy_leg_only.py --only ...
CS_DEFAULT_KERNEL = "ngm::cs_canon_kernel<3, 6, 2, 1, 7, true>"
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


def make_reads(contigs, n, seed, paired=False, subs=0.01, indel_bases=0.0, starts=None):
    """uniform positions, 50 % reverse strand, 1 % substitutions, one 1-3 bp indel in 15 % of the reads (0.1 % of the
    bases).  paired: rows 2i / 2i+1 are the two ends of a fragment, insert size ~ N(350, 35), FR orientation, half of
    the fragments from the reverse strand.  Returns ([n, Q] uint8 rows, truth contig, truth pos)."""
    rng = np.random.default_rng(seed)
    lens = np.array([len(c) for c in contigs], dtype=np.float64)
    rows = np.zeros((n, Q), np.uint8)
    pos = np.zeros(n, np.int64)
    if paired:
        nf = n // 2
        cf = rng.choice(len(contigs), size=nf, p=lens / lens.sum()) if starts is None else starts[0]
        ins = np.maximum(READ_LEN + 10, rng.normal(350, 35, nf).astype(np.int64))
        ci = np.repeat(cf, 2)
        flip = rng.random(nf) < 0.5
        for k in range(len(contigs)):
            sel = np.nonzero(cf == k)[0]
            if sel.size == 0:
                continue
            p = rng.integers(0, len(contigs[k]) - 600, sel.size) if starts is None else np.minimum(starts[1][sel], len(contigs[k]) - 600)
            left = contigs[k][p[:, None] + np.arange(READ_LEN)[None, :]]
            pr = p + ins[sel] - READ_LEN
            right = COMP[contigs[k][pr[:, None] + np.arange(READ_LEN)[None, :]][:, ::-1]]
            f = flip[sel]
            rows[2 * sel, :READ_LEN] = np.where(f[:, None], right, left)
            rows[2 * sel + 1, :READ_LEN] = np.where(f[:, None], left, right)
            pos[2 * sel] = np.where(f, pr, p)
            pos[2 * sel + 1] = np.where(f, p, pr)
    else:
        ci = rng.choice(len(contigs), size=n, p=lens / lens.sum()) if starts is None else starts[0]
        for k in range(len(contigs)):
            sel = np.nonzero(ci == k)[0]
            if sel.size == 0:
                continue
            p = rng.integers(0, len(contigs[k]) - READ_LEN - 8, sel.size) if starts is None else np.minimum(starts[1][sel], len(contigs[k]) - READ_LEN - 8)
            pos[sel] = p
            rows[sel, :READ_LEN] = contigs[k][p[:, None] + np.arange(READ_LEN)[None, :]]
    sub = rng.random((n, READ_LEN)) < subs
    rows[:, :READ_LEN][sub] = ACGT[rng.integers(0, 4, int(sub.sum()))]
    # indels: one 1-3 bp indel in 15 % of the reads (0.1 % of the bases); with indel_bases > 0 (config #5: 3 %) every read gets
    # READ_LEN * indel_bases / 2 more of them
    events = np.nonzero(rng.random(n) < 0.15)[0]
    if indel_bases > 0:
        events = np.concatenate([events, np.repeat(np.arange(n), max(1, int(round(READ_LEN * indel_bases / 2.0))))])
    for i in events:
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


def write_fastq(rows, paths):
    """rows -> plain 4-line FASTQ with fixed-width names (numpy, no Python loop).  Two paths: rows 2i / 2i+1 are mates and go to
    paths[0] / paths[1] as r<8 digits>/1 and /2; one path: single-end reads r<8 digits>.  Returns the bytes per record."""
    two = len(paths) == 2
    n = rows.shape[0] // 2 if two else rows.shape[0]
    idx = np.arange(n, dtype=np.int64)
    rec = 0
    for mate, path in enumerate(paths):
        w = 12 if two else 10
        hdr = np.empty((n, w), np.uint8)
        hdr[:, 0] = ord("@"); hdr[:, 1] = ord("r")
        for d in range(8):
            hdr[:, 2 + d] = ord("0") + (idx // 10 ** (7 - d)) % 10
        if two:
            hdr[:, 10] = ord("/"); hdr[:, 11] = ord("1") + mate
        nl = np.full((n, 1), ord("\n"), np.uint8)
        body = rows[mate::2, :READ_LEN] if two else rows[:, :READ_LEN]
        a = np.concatenate([hdr, nl, body, np.frombuffer(b"\n+\n", np.uint8)[None, :].repeat(n, 0),
                            np.full((n, READ_LEN), ord("I"), np.uint8), nl], axis=1)
        a.tofile(path)
        rec = a.shape[1]
    return rec


def _sam_body(path, limit=None):
    out = {}
    with open(path, "rb") as f:
        for line in f:
            if line[:1] == b"@":
                continue
            if limit is not None and len(out) >= limit:
                break
            t = line.split(b"\t", 2)
            out[(t[0], int(t[1]) & 0xC0)] = line
    return out


SAM_FIELDS = ["QNAME", "FLAG", "RNAME", "POS", "MAPQ", "CIGAR", "RNEXT", "PNEXT", "TLEN", "SEQ", "QUAL"]


def _sam_diff(theirs, ours, limit=4):
    """(identical lines, the first differing records as {name, field: [reference, ours]})"""
    same, diffs = 0, []
    for k, v in theirs.items():
        o = ours.get(k)
        if o == v:
            same += 1
            continue
        if len(diffs) < limit:
            if o is None:
                diffs.append({"name": k[0].decode(), "missing_in_ours": True})
                continue
            fa, fb = v.decode().rstrip("\n").split("\t"), o.decode().rstrip("\n").split("\t")
            d = {"name": k[0].decode(), "mapq_reference": fa[4], "mapq_ours": fb[4]}
            for i in range(max(len(fa), len(fb))):
                x, y = (fa[i] if i < len(fa) else None), (fb[i] if i < len(fb) else None)
                if x != y:
                    d[SAM_FIELDS[i] if i < len(SAM_FIELDS) else (x or y)[:2]] = [x, y]
            diffs.append(d)
    return same, diffs


def usable_cpus():
    """CPUs this process may use at once: os.cpu_count() capped by the cgroup's CFS quota (the GPU boxes of this pool: 256 hardware
    threads, cpu.max = 16 CPUs; more runnable threads than the quota get LESS done, profiles/r04_cpu_quota_probe.txt)"""
    n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(per))))
    except Exception:
        pass
    return n


def end_to_end(ref, contigs, workdir, args, paired, affine, sens):
    """The drop-in, measured: `ngm-hip` from the first input byte to the closed SAM file (plain FASTQ, .fastq.gz and --bam), and
    NextGenMap itself (-t cores and -t 1) on slices of the same input with the SAM records compared field by field."""
    import re
    import ref_files as RF
    from nextgenmap_amd import build as B
    cores = usable_cpus()
    fa = os.path.join(workdir, "bench_ref.fa")
    if not os.path.exists(fa):
        with open(fa, "w") as f:
            f.write(">stub\nACGT\n")  # with the caches present both programs only check that the file exists
        ref.write_ngm_cache(fa)
    n = args.e2e_reads & ~1
    t0 = time.perf_counter()
    rows, _, _ = make_reads(contigs, n, seed=20240602 + 3, paired=paired, subs=args.subs, indel_bases=args.indel_bases)  # config #3's seed
    files = [os.path.join(workdir, "e2e_1.fq"), os.path.join(workdir, "e2e_2.fq")] if paired else [os.path.join(workdir, "e2e.fq")]
    rec = write_fastq(rows, files)
    t_make = time.perf_counter() - t0
    common = ["-s", "%.6f" % sens, "--no-progress", "--max-read-length", str(READ_LEN)] + (["--affine"] if affine else [])
    if args.corridor > 0:
        common += ["-C", str(args.corridor // 2)]

    def inputs(fs):
        return ["-1", fs[0], "-2", fs[1]] if paired else ["-q", fs[0]]

    def run_hip(fs, outp, extra=()):
        cmd = [B.CLI, "-r", fa] + inputs(fs) + ["-o", outp] + common + list(extra)
        t = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True)
        wall = time.perf_counter() - t
        log = r.stdout + r.stderr
        if r.returncode != 0 or "Done" not in log:
            raise RuntimeError("ngm-hip failed: " + log[-600:])
        m = re.search(r"Input to output: ([0-9.]+) s", log)
        mi = re.search(r"Reference and index ready: ([0-9.]+) s", log)
        mg = re.search(r"GPU kernels: ([0-9.]+) s of the ([0-9.]+) s mapping pass", log)
        return {"cmd": cmd, "log": log, "wall": wall, "io": float(m.group(1)) if m else wall, "index": float(mi.group(1)) if mi else None,
                "gpu": float(mg.group(1)) if mg else None, "pass": float(mg.group(2)) if mg else None}

    sam = os.path.join(workdir, "e2e.sam")
    # (one sample per box is not a measurement) the same run --e2e-runs times: the first one's SAM is the one compared below;
    # between runs the file is removed and the dirty pages of the last run are flushed (a run that writes 4 GB leaves 4 GB of write-back
    # behind, and the next run's writes wait for it: measured in round 4, mapping pass 0.50 s -> 0.89 s)
    runs = []
    h = None
    for rep in range(max(1, args.e2e_runs)):
        if rep > 0:
            os.remove(sam)
            os.sync()
        hr = run_hip(files, sam)
        runs.append(hr)
        if rep == 0:
            h = hr
    ios = sorted(r_["io"] for r_ in runs)
    med = ios[len(ios) // 2]
    h_med = [r_ for r_ in runs if r_["io"] == med][0]
    out = {"reads": n, "seconds_first_input_byte_to_sam_closed": med, "value": n / med, "unit": "reads/s",
           "runs": {"n": len(runs), "seconds_min_median_max": [ios[0], med, ios[-1]], "reads_per_s_min_median_max": [n / ios[-1], n / med, n / ios[0]],
                    "per_run": [{"input_to_output_s": r_["io"], "mapping_pass_s": r_["pass"], "gpu_kernel_s": r_["gpu"], "index_load_s": r_["index"], "process_wall_s": r_["wall"]} for r_ in runs],
                    "note": "value = the median run; the output file is removed and os.sync() called between runs"},
           "index_load_s": h_med["index"], "process_wall_s": h_med["wall"], "reads_per_s_of_process_wall": n / h_med["wall"],
           "sam_bytes": os.path.getsize(sam), "fastq_bytes": len(files) * (n // len(files)) * rec,
           "command": " ".join(["ngm-hip"] + h["cmd"][1:]), "cli_log_tail": [l for l in h["log"].splitlines() if "Before the mapping pass" in l or "Sensitivity estimate, s" in l] + [l for l in h["log"].splitlines() if "MAIN" in l][-4:],
           "input": "%s plain FASTQ (%d x %d bp %s, fixed-width names), page cache warm; index from NextGenMap cache files"
                    % ("two" if paired else "one", n // 2 if paired else n, READ_LEN, "pairs" if paired else "reads"),
           "make_input_s": t_make, "gpu_kernel_s": h_med["gpu"], "mapping_pass_s": h_med["pass"],
           "gpu_busy_fraction_of_mapping_pass": (h_med["gpu"] / h_med["pass"]) if h_med["gpu"] and h_med["pass"] else None}
    # the same program on what real input looks like (.fastq.gz: the serial reader) and with --bam, on a slice
    ng = min(n, args.e2e_gz_reads) & ~1
    # what the later comparisons need of the big SAM file stays in memory (the first records, in file order); the file itself goes now:
    # 4 GB of dirty page cache behind the runs that follow make their writes wait for the write-back (measured with the same input
    # written twice: mapping pass 0.50 s into a fresh file system state, 0.89 s behind the first run's file)
    keep = min(n, max(ng, args.cpu_sample_reads, args.cpu_t1_reads))
    ours = _sam_body(sam, keep)
    out["sam_bytes"] = os.path.getsize(sam)
    os.remove(sam)
    if ng > 0:
        try:
            cnt = ng // len(files)
            slices = []
            for i, src in enumerate(files):
                dst = os.path.join(workdir, "gz_%d.fq" % i)
                with open(src, "rb") as fi, open(dst, "wb") as fo:
                    fo.write(fi.read(cnt * rec))
                slices.append(dst)
            hb = run_hip(slices, os.path.join(workdir, "e2e.bam"), ["--bam"])
            out["bam_output"] = {"reads": ng, "value": ng / hb["io"], "unit": "reads/s", "seconds": hb["io"], "bam_bytes": os.path.getsize(os.path.join(workdir, "e2e.bam"))}
            t = time.perf_counter()
            for dst in slices:
                subprocess.run(["gzip", "-1", "-f", dst], check=True)
            t_gz = time.perf_counter() - t
            hz = run_hip([d + ".gz" for d in slices], os.path.join(workdir, "e2e_gz.sam"))
            # the first ng records of the plain run (same order, same batches) against the .gz run's
            with open(os.path.join(workdir, "e2e_gz.sam"), "rb") as fb_:
                rb = [l for l in fb_ if l[:1] != b"@"]
            same = len(rb) == ng and all(x == y for x, y in zip(ours.values(), rb))
            out["fastq_gz_input"] = {"reads": ng, "value": ng / hz["io"], "unit": "reads/s", "seconds": hz["io"], "gzip_1_of_the_input_s": t_gz,
                                     "same_sam_as_plain_input": same}
            for fn in [d + ".gz" for d in slices] + [os.path.join(workdir, "e2e.bam"), os.path.join(workdir, "e2e_gz.sam")]:
                os.remove(fn)
        except Exception as e:
            out["variants_error"] = str(e)[:300]
    base = None
    if not args.no_cpu_baseline and RF.have_reference_binary() and affine:
        threads = min(os.cpu_count() or 1, 64)   # (the reference does best with more threads than the quota allows CPUs: 114 k reads/s at -t 64 against 99 k at -t 16)

        def slice_to(cnt_reads, tag):
            cnt = cnt_reads // len(files)
            dsts = []
            for i, src in enumerate(files):
                dst = os.path.join(workdir, "%s_%d.fq" % (tag, i))
                with open(src, "rb") as fi, open(dst, "wb") as fo:
                    fo.write(fi.read(cnt * rec))
                dsts.append(dst)
            return dsts

        def run_ref(fs, outp, t):
            c = [RF.NGM_CORE, "-r", fa] + inputs(fs) + ["-o", outp, "-t", str(t)] + common
            t0_ = time.perf_counter()
            rr = subprocess.run(c, capture_output=True, text=True, cwd=workdir)
            dt = time.perf_counter() - t0_
            if "Done" not in (rr.stdout + rr.stderr):
                raise RuntimeError("reference run failed: " + (rr.stdout + rr.stderr)[-400:])
            return dt
        t_load = run_ref(slice_to(2, "one"), os.path.join(workdir, "one.sam"), threads)
        # the sample: what the reference maps in ~25 s on this host (a pilot of 20 000 reads gives the rate), at most --cpu-sample-reads
        pilot = min(20000, n) & ~1
        t_pilot = max(run_ref(slice_to(pilot, "pilot"), os.path.join(workdir, "pilot.sam"), threads) - t_load, 1e-3)
        ns = int(min(args.cpu_sample_reads, n, max(pilot, 25.0 * pilot / t_pilot))) & ~1
        ref_sam = os.path.join(workdir, "ref.sam")
        t_all = run_ref(slice_to(ns, "s"), ref_sam, threads)
        t_map = max(t_all - t_load, 1e-3)
        ref_body = _sam_body(ref_sam)
        same, diffs = _sam_diff(ref_body, ours)
        # the same reference command a second time: which of its own lines move between two -t N runs (thread scheduling decides which reads
        # share a CS thread's running mean insert size), and are the lines where ngm-hip differs among the reads it is undecided on itself?
        self_check = None
        if not args.no_reference_rerun:
            ref2_sam = os.path.join(workdir, "ref2.sam")
            run_ref(slice_to(ns, "s"), ref2_sam, threads)
            ref2_body = _sam_body(ref2_sam)
            same2, diffs2 = _sam_diff(ref_body, ref2_body)
            moved = {k_ for k_ in ref_body if ref2_body.get(k_) != ref_body[k_]}
            ours_off = {k_ for k_ in ref_body if ours.get(k_) != ref_body[k_]}
            self_check = {"records_compared": len(ref_body), "identical_lines_between_two_reference_runs": same2, "first_differences": diffs2[:3],
                          "ngm_hip_lines_that_differ_from_run_1": len(ours_off),
                          "of_these_equal_to_run_2_or_moved_between_the_runs": len({k_ for k_ in ours_off if k_ in moved or ours.get(k_) == ref2_body.get(k_)})}
            os.remove(ref2_sam)
        # ... and -t 1, the run whose output this library reproduces exactly (one CS thread: one running mean insert size)
        n1 = min(args.cpu_t1_reads, n) & ~1
        th1, same1, diffs1, t_t1, early = {}, 0, [], 0.0, None
        if n1 > 0:
            t1_sam = os.path.join(workdir, "ref_t1.sam")
            t1_files = slice_to(n1, "t1")
            t_t1 = run_ref(t1_files, t1_sam, 1)
            th1 = _sam_body(t1_sam)
            same1, diffs1 = _sam_diff(th1, ours)
            # ngm-hip on the same slice: the pairs it loses because the reference loses them (first mate fills the score buffer exactly, second mate
            # without candidates: DESIGN.md 2)
            try:
                early = [l for l in run_hip(t1_files, os.path.join(workdir, "ours_t1.sam"))["log"].splitlines() if "Pairs lost as NextGenMap" in l][-1:]
                os.remove(os.path.join(workdir, "ours_t1.sam"))
            except Exception as e:
                early = [str(e)[:200]]
        # the reference against itself across thread counts: the same reads in its -t N and its -t 1 output
        t1_vs_tn = None
        if th1:
            both = [k_ for k_ in th1 if k_ in ref_body]
            ref_moves = [k_ for k_ in both if th1[k_] != ref_body[k_]]
            ours_off_here = [k_ for k_ in both if ours.get(k_) != ref_body[k_]]
            t1_vs_tn = {"records_in_both_runs": len(both), "lines_the_reference_writes_differently_at_t1_and_tN": len(ref_moves),
                        "ngm_hip_lines_that_differ_from_tN_here": len(ours_off_here),
                        "of_these_identical_to_the_reference_at_t1": len([k_ for k_ in ours_off_here if ours.get(k_) == th1[k_]])}
        base = {"value": ns / t_map, "unit": "reads/s", "cores": threads, "cpu_quota": cores, "kind": "reference",
                "sample": "NextGenMap 0.5.5 ngm-core --affine -t %d (the CPUs this container may use: %d hardware threads, cgroup quota %d) on the first %d reads of the "
                          "end-to-end input vs the same genome (index loaded from the same cache files): %.1f s total minus %.1f s index load/start-up measured "
                          "with a 1-pair run" % (threads, os.cpu_count() or 1, cores, ns, t_all, t_load),
                "parity_vs_reference_sam": {"records_compared": ns, "identical_lines": same, "first_differences": diffs,
                                            "note": "whole SAM lines, differing fields listed; the reference ran %d CS threads (ngm-hip reproduces its -t 1 output: "
                                                    "parity_vs_reference_sam_t1).  reference_vs_itself says which lines the reference's own two -t %d runs moved and how many of "
                                                    "the lines where ngm-hip differs are among them or equal run 2; the others are unexplained by this run" % (threads, threads),
                                            "reference_vs_itself": self_check, "reference_t1_vs_tN": t1_vs_tn},
                "parity_vs_reference_sam_t1": {"records_compared": len(th1), "identical_lines": same1, "first_differences": diffs1, "seconds": t_t1, "ngm_hip_on_the_same_slice": early,
                                               "note": "ngm-core --affine -t 1 on the first %d reads: the run ngm-hip reproduces" % n1}}
    for fn in files:
        try:
            os.remove(fn)
        except OSError:
            pass
    return out, base


def sharded_end_to_end(workdir, contigs, args, paired, affine, sens, world, exe):
    """BASELINE config #4 as the product runs it (N > 1, rank 0 only, the other ranks wait at a barrier): ONE input of --e2e-reads reads
    through `ngm-hip -g 0,...,N-1 --shard-output` -- one process per GPU, every shard its own reference copy (from the cache files rank 0
    wrote), its own slice of the record index, its own output file, appended in shard order -- from the first input byte to the closed,
    concatenated SAM file.  "strong" scaling: the job is fixed, the GPUs split it."""
    import re
    fa = os.path.join(workdir, "bench_ref.fa")
    n = args.e2e_reads & ~1
    t0 = time.perf_counter()
    rows, _, _ = make_reads(contigs, n, seed=20240602 + 3, paired=paired, subs=args.subs, indel_bases=args.indel_bases)
    files = [os.path.join(workdir, "e2e_1.fq"), os.path.join(workdir, "e2e_2.fq")] if paired else [os.path.join(workdir, "e2e.fq")]
    write_fastq(rows, files)
    del rows
    t_make = time.perf_counter() - t0
    sam = os.path.join(workdir, "e2e_sharded.sam")
    cmd = [exe, "-r", fa] + (["-1", files[0], "-2", files[1]] if paired else ["-q", files[0]]) + ["-o", sam, "-s", "%.6f" % sens, "--no-progress",
           "--max-read-length", str(READ_LEN), "-g", ",".join(str(g) for g in range(world)), "--shard-output"] + (["--affine"] if affine else [])
    if args.corridor > 0:
        cmd += ["-C", str(args.corridor // 2)]
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True)
    wall = time.perf_counter() - t0
    log = r.stdout + r.stderr
    if r.returncode != 0:
        raise RuntimeError("sharded ngm-hip failed: " + log[-600:])
    io = [float(x) for x in re.findall(r"Input to output: ([0-9.]+) s", log)]
    idx = [float(x) for x in re.findall(r"Reference and index ready: ([0-9.]+) s", log)]
    done = [(int(a), int(b), int(c)) for a, b, c in re.findall(r"Done \((\d+) reads mapped \([0-9.]+%\), (\d+) reads not mapped, (\d+) lines written\)", log)]
    app = re.search(r"shards appended to the output in ([0-9.]+) s", log)
    lines = 0
    with open(sam, "rb") as f:
        for l in f:
            lines += l[:1] != b"@"
    # the product's own reduction (round 5): the parent sums the int64[8] vectors its shard processes hand over and prints ONE line; on
    # distinct GPUs shard 0 also prints what the ncclAllReduce among the shards gave (the per-shard lines stay as a cross-check)
    summed = re.search(r"Done, (\d+) shards summed \((\d+) reads mapped \([0-9.]+%\), (\d+) reads not mapped, (\d+) lines written; (\d+) reads; (\d+) pairs with both mates mapped, (\d+) of them broken, mean insert size ([0-9.]+)\)", log)
    rccl = re.search(r"Statistics all-reduce over (\d+) GPUs \(RCCL, (\d+) us\): (\d+) reads, (\d+) mapped, (\d+) not mapped, (\d+) lines written", log)
    out = {"reads": n, "shards": world, "scaling": "strong", "unit": "reads/s",
           "stats_summed_by_the_parent_process": ({"shards": int(summed.group(1)), "mapped": int(summed.group(2)), "unmapped": int(summed.group(3)), "written": int(summed.group(4)),
                                                   "reads": int(summed.group(5)), "pairs_total": int(summed.group(6)), "pairs_broken": int(summed.group(7)), "mean_insert_size": float(summed.group(8))}
                                                  if summed else None),
           "stats_allreduce_rccl_among_the_shards": ({"ranks": int(rccl.group(1)), "microseconds": int(rccl.group(2)), "reads": int(rccl.group(3)), "mapped": int(rccl.group(4)),
                                                      "unmapped": int(rccl.group(5)), "written": int(rccl.group(6))} if rccl else None),
           "seconds_first_input_byte_to_concatenated_sam_closed": (max(io) if io else wall) + (float(app.group(1)) if app else 0.0),
           "per_shard_input_to_output_s": io, "per_shard_index_load_s": idx, "append_s": float(app.group(1)) if app else None, "process_wall_s": wall,
           "stats_summed_over_shards": {"mapped": sum(d[0] for d in done), "unmapped": sum(d[1] for d in done), "written": sum(d[2] for d in done)},
           "sam_records": lines, "sam_bytes": os.path.getsize(sam), "make_input_s": t_make, "command": " ".join(["ngm-hip"] + cmd[1:])}
    out["value"] = n / out["seconds_first_input_byte_to_concatenated_sam_closed"]
    out["reads_per_s_of_process_wall"] = n / wall
    for fn in files:
        try:
            os.remove(fn)
        except OSError:
            pass
    return out


def heavy_tail_leg(args, dev, local_rank, paired, affine, sens, workdir):
    """The second headline: the same resident mapping path on a genome with a GRCh38-LIKE k-mer spectrum (tests/humanlike.py: one
    SINE-like family at ~10^5 copies per 300 Mbp, LINE-like families, satellite arrays, microsatellites, segmental duplications,
    isochores) of GRCh38's size.  Two sub-legs, each over --read-sets distinct read sets that the steps rotate through, each with its
    own `roofline` (dominant kernel of THAT workload, SURVEY.md 8(d) bytes, HIP events) and `cpu_baseline` (ngm-core --affine on a
    slice of the same reads against the same genome, its SAM compared with ngm-hip's on that slice):
      reads_drawn_uniformly            fragments start anywhere (what sequencing a genome gives): the realistic one
      half_of_the_reads_from_repeats   half of the fragments start inside a repeat instance (kinds equally likely): the stress"""
    import re
    import threading
    import torch
    import humanlike as HL
    import ref_files as RF
    from nextgenmap_amd import build as B
    from nextgenmap_amd.pipeline import HIT_DTYPE, Mapper, Reference
    t0 = time.perf_counter()
    G = HL.make_genome(total_bp=int(args.heavy_tail_mbp * 1e6), n_contigs=24, seed=20260929)
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    ref = Reference.from_contigs(G.contigs, device=local_rank, kmer=KMER, kmer_skip=2, bin_size=2)
    t_index = time.perf_counter() - t0
    R = args.reads_per_step
    S = max(1, min(args.read_sets, args.heavy_tail_steps))
    band = C + 1 if affine else C
    # (this workload's host stages -- order replay, the sequential part of the pair selection -- outweigh the uniform genome's: four mapper
    # instances per GPU instead of two keep the GPU fed; round 4, one box, 2 / 3 / 4 instances: 1.13 / 1.41 / 1.47 M reads/s)
    W = max(1, min(max(args.workers, args.heavy_tail_workers), R // 2048))
    bounds = [(R * w // W) & ~1 for w in range(W)] + [R]
    kw = dict(gap_read=33, gap_ref=33, gap_extend=3, personality=1) if affine else {}
    mps = [Mapper(ref, Q, C, sensitivity=sens, **kw) for _ in range(W)]
    with_cpu = (not args.no_cpu_baseline) and affine and RF.have_reference_binary() and args.heavy_tail_cpu_reads > 0
    cores = usable_cpus()
    threads = min(os.cpu_count() or 1, 64)
    fa = os.path.join(workdir, "heavy_ref.fa")
    t_cache = t_load = 0.0
    common = ["-s", "%.6f" % sens, "--no-progress", "--max-read-length", str(READ_LEN)] + (["--affine"] if affine else [])

    def inputs(fs):
        return ["-1", fs[0], "-2", fs[1]] if paired else ["-q", fs[0]]

    def run_ref(fs, outp, t):
        c = [RF.NGM_CORE, "-r", fa] + inputs(fs) + ["-o", outp, "-t", str(t)] + common
        t0_ = time.perf_counter()
        rr = subprocess.run(c, capture_output=True, text=True, cwd=workdir)
        dt = time.perf_counter() - t0_
        if "Done" not in (rr.stdout + rr.stderr):
            raise RuntimeError("reference run failed: " + (rr.stdout + rr.stderr)[-400:])
        return dt

    def slice_files(rows, n, tag):
        fs = [os.path.join(workdir, "%s_1.fq" % tag), os.path.join(workdir, "%s_2.fq" % tag)] if paired else [os.path.join(workdir, "%s.fq" % tag)]
        write_fastq(rows[:n], fs)
        return fs
    cpu_err = None
    with_e2e = args.heavy_tail_e2e_reads > 0 and affine
    if with_cpu or with_e2e:
        try:
            t0 = time.perf_counter()
            with open(fa, "w") as f:
                f.write(">stub\nACGT\n")  # with the caches present both programs only check that the file exists
            ref.write_ngm_cache(fa)
            t_cache = time.perf_counter() - t0
        except Exception as e:
            with_cpu, with_e2e, cpu_err = False, False, str(e)[:300]

    kept, kept_sets = {}, {}   # per sub-leg: the reference's SAM of the cpu_baseline slice, the read sets (the end-to-end run below starts with them)

    def sub_leg(tag, share, seed0):
        nonlocal t_load
        t0 = time.perf_counter()
        sets = []
        for s_ in range(S):
            starts = HL.sample_starts(G, R // 2 if paired else R, 400 if paired else READ_LEN, seed=seed0 + 17 * s_, repeat_share=share)
            sets.append(make_reads(G.contigs, R, seed=seed0 + 1 + 17 * s_, paired=paired, subs=args.subs, indel_bases=args.indel_bases, starts=starts))
        d_sets = [torch.from_numpy(t_[0]).to(dev) for t_ in sets]
        t_reads = time.perf_counter() - t0
        out = (np.zeros(R, HIT_DTYPE), np.zeros((R, 4 * Q), np.uint8), np.zeros((R, 4 * Q), np.uint8))
        kms = [np.zeros(9) for _ in range(W)]

        def worker(w, steps, first):
            lo, hi = bounds[w], bounds[w + 1]
            for i_ in range(steps):
                s_ = (first + i_) % S
                (mps[w].map_pe_raw if paired else mps[w].map_se_raw)(sets[s_][0][lo:hi], d_sets[s_][lo:hi], tuple(o[lo:hi] for o in out))
                kms[w][:8] += np.array(mps[w].last_kernel_ms())
                kms[w][8] += mps[w].last_order_replay_ms()

        def run(steps, first=0):
            ts = [threading.Thread(target=worker, args=(w, steps, first)) for w in range(W)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
        run(S)   # set-up: every read set once (buffers grow to their largest batch, the running mean insert size settles)
        for k in kms:
            k[:] = 0
        before = [m_.path_counters() for m_ in mps]
        before_t = sum(m_.order_table_reads() for m_ in mps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(args.heavy_tail_steps)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        steps = args.heavy_tail_steps
        pc = {k: sum(m_.path_counters()[k] - b[k] for m_, b in zip(mps, before)) for k in before[0]}
        ctr = np.sum([m_.cs_counters() for m_ in mps], axis=0)   # of the last step
        km = np.sum(kms, axis=0) / steps
        rows_last, truth_c, truth_p = sets[(steps - 1) % S]
        hits = out[0]
        mapped = hits["mapped"] == 1
        correct = mapped & (hits["contig"] == truth_c) & (np.abs(hits["pos"].astype(np.int64) - truth_p) <= C // 2 + 4)
        kmers, hits_voted, n_cand, n_aln = int(ctr[0]), int(ctr[1]), int(ctr[2]), int(mapped.sum())
        b_cs = 20 * kmers + 4 * hits_voted + 16 * n_cand
        kernels = {"candidate_search": (km[0], b_cs, "cs_fast_kernel / cs_heavy2_kernel / cs_global_kernel (candidate search): 20 B/k-mer + 4 B/index hit + 16 B/candidate"),
                   "sw_score": (km[2], n_cand * (Q + Q + C + 4), "sw_*score*_kernel (BatchScore): B_score = q + (q + c) + 4 bytes per pair"),
                   "sw_align": (km[5] + km[6], n_aln * (Q + Q + C + 8 + 4 * (2 * Q + C + 1)), "sw_*align*_kernel + traceback (BatchAlign): B_align = q + (q + c) + 8 + 4 (2q + c + 1) bytes per pair")}
        dom = max(kernels, key=lambda k_: kernels[k_][0])

        def roof(name):
            k_ms, k_bytes, k_note = kernels[name]
            achieved = k_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
            traffic = traffic_source = None
            if name == "candidate_search":
                fn = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", HEAVY_CS_TRAFFIC_FILE % ("uniform" if share == 0.0 else "repeats"))
                try:
                    with open(fn) as f:
                        tj = json.load(f)
                    if R == 1 << 20:   # (the PMC pass's step: HEAVY_CS_TRAFFIC_SEARCHES searches of 1 048 576 reads in all -- the bytes do not depend on how a step is cut)
                        traffic = int(tj["candidate_search_bytes_per_batch"]) * HEAVY_CS_TRAFFIC_SEARCHES
                        traffic_source = "profiles/" + os.path.basename(fn) + " [candidate_search_bytes_per_batch x %d searches per step of the PMC pass]" % HEAVY_CS_TRAFFIC_SEARCHES
                except (OSError, ValueError, KeyError):
                    pass
            return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                    "traffic_over_algorithmic": (traffic / k_bytes) if traffic and k_bytes else None,
                    "kernel": name, "kernel_note": k_note, "kernel_ms_per_step": float(k_ms), "bytes_per_step": int(k_bytes)}
        roofline = roof(dom)
        roofline["note"] = ("the kernel group with the largest GPU time per step of THIS workload; achieved = its algorithmic bytes (SURVEY.md 8d) / its HIP-event "
                            "time on the launch streams (summed over the mapper instances); traffic = HBM bytes per step of the same group from this round's committed "
                            "rocprofv3 PMC passes of this sub-leg (FETCH_SIZE / WRITE_SIZE in separate passes, profiles/summarize_rocprof.py), per search x the "
                            "searches of a step -- not measured in this run"
                            + ("" if dom == "candidate_search" else "; the score kernels are VALU-bound (sw_gcells_per_s), candidate search is priced beside it under `candidate_search`"))
        if dom != "candidate_search":
            roofline["candidate_search"] = roof("candidate_search")
        nr = max(1, pc["reads"])
        ms_step = elapsed / steps * 1e3
        all_k = float(km[:7].sum())
        res = {"value": R * steps / elapsed, "unit": "reads/s", "steps": steps, "ms_per_step": ms_step, "read_sets": S,
               "reads": "%d x %d bp %s per step, %.0f %% of the fragments start inside a repeat instance (kinds equally likely), %.1f %% substitutions; the steps rotate through %d distinct read sets"
                        % (R, READ_LEN, "PE" if paired else "SE", 100 * share, 100 * args.subs, S),
               "per_read": {"candidates": n_cand / R, "index_hits": hits_voted / R, "kmers": kmers / R},
               "share_of_reads": {"heavy_read_kernel": pc["heavy"] / nr, "exact_search_lds_table": pc["exact_lds"] / nr, "exact_search_global_table": pc["exact_global"] / nr,
                                  "candidate_order_replayed": pc["order_replayed"] / nr, "candidate_order_exact_global_replay": pc["order_exact_global"] / nr,
                                  "candidate_order_replay_with_a_table_in_global_memory": (sum(m_.order_table_reads() for m_ in mps) - before_t) / nr,
                                  "candidate_order_undetermined": pc["order_undetermined"] / nr},
               "kernel_ms": {"candidate_search": km[0], "gather_score": km[1], "sw_score": km[2], "select": km[3], "gather_align": km[4], "sw_align": km[5], "traceback": km[6],
                             "all_kernels": all_k, "candidate_order_replay_on_its_own_stream": km[8], "candidate_search_stage_incl_host_sync": km[7]},
               "gpu_kernels_fraction_of_step": {"stage_kernels": all_k / ms_step, "with_order_replay": (all_k + km[8]) / ms_step},
               "sw_gcells_per_s": {"score_kernel": n_cand * READ_LEN * band / (km[2] * 1e-3) / 1e9 if km[2] > 0 else None,
                                   "align_kernel": n_aln * READ_LEN * band / (km[5] * 1e-3) / 1e9 if km[5] > 0 else None},
               "roofline": roofline,
               "accuracy": {"mapped": float(mapped.mean()), "within_band_of_truth": float(correct.mean()), "mapq_gt0": float((hits["mapq"] > 0).mean())},
               "setup_s": {"read_sets": t_reads}}
        if with_cpu:
            try:
                if t_load == 0.0:
                    t_load = run_ref(slice_files(sets[0][0], 2, "h_one"), os.path.join(workdir, "h_one.sam"), threads)
                # (a sample large enough that the paths only this genome takes are in it -- 50 000 records by default, 1-2.5
                # minutes of the reference on a 16-CPU quota -- instead of what it maps in ~8 s)
                ns = int(min(args.heavy_tail_cpu_reads, R)) & ~1
                fs = slice_files(sets[0][0], ns, "h_" + tag[:4])
                ref_sam, hip_sam = os.path.join(workdir, "h_ref.sam"), os.path.join(workdir, "h_hip.sam")
                t_all = run_ref(fs, ref_sam, threads)
                t_map = max(t_all - t_load, 1e-3)
                cmd = [B.CLI, "-r", fa] + inputs(fs) + ["-o", hip_sam] + common
                t1 = time.perf_counter()
                rr = subprocess.run(cmd, capture_output=True, text=True)
                t_hip = time.perf_counter() - t1
                if rr.returncode != 0:
                    raise RuntimeError("ngm-hip failed: " + (rr.stdout + rr.stderr)[-400:])
                mio = re.search(r"Input to output: ([0-9.]+) s", rr.stdout + rr.stderr)
                ref_body = _sam_body(ref_sam)
                kept[tag] = (ns, ref_body)
                same, diffs = _sam_diff(ref_body, _sam_body(hip_sam))
                res["cpu_baseline"] = {"value": ns / t_map, "unit": "reads/s", "cores": threads, "cpu_quota": cores, "kind": "reference",
                                       "sample": "NextGenMap 0.5.5 ngm-core --affine -t %d (%d hardware threads visible, cgroup quota %d CPUs) on the first %d reads of this sub-leg's "
                                                 "read set 0 vs the same genome (index from the cache files this library wrote): %.1f s total minus %.1f s index load / start-up "
                                                 "measured with a 1-pair run" % (threads, os.cpu_count() or 1, cores, ns, t_all, t_load),
                                       "parity_vs_reference_sam": {"records_compared": ns, "identical_lines": same, "first_differences": diffs,
                                                                   "note": "whole SAM lines; the reference runs %d CS threads, each with its own running mean insert size "
                                                                           "(ScoreBuffer.h:90): equal-score pair ties may differ from its -t 1 output, which is what ngm-hip reproduces "
                                                                           "(tests/test_gpu_humanlike.py; profiles/r05_humanlike_t1_2M_reads.log)" % threads},
                                       "ngm_hip_on_the_same_slice": {"seconds_first_input_byte_to_sam_closed": float(mio.group(1)) if mio else None, "process_wall_s": t_hip}}
                for fn in fs + [ref_sam, hip_sam]:
                    try:
                        os.remove(fn)
                    except OSError:
                        pass
            except Exception as e:
                res["cpu_baseline"] = {"error": str(e)[:400]}
        elif cpu_err:
            res["cpu_baseline"] = {"error": cpu_err}
        del d_sets
        kept_sets[tag] = [t_[0] for t_ in sets]
        return res

    out = {"genome": "tests/humanlike.py make_genome(%d Mbp, 24 contigs, seed 20260929): %d repeat instances; automatic max. k-mer frequency %d (uniform genome: 100)"
                     % (args.heavy_tail_mbp, len(G.repeats), ref.auto_max_kfreq),
           "mapper_instances_per_gpu": W,
           "setup_s": {"genome_generation": t_gen, "encode+index_build": t_index, "index_entries": ref.index_entries, "cache_files_for_the_reference_program": t_cache}}
    only = getattr(args, "heavy_tail_only", "both")   # (profiles/tools/heavy_leg_only.py --only: one sub-leg, for the PMC passes)
    if only in ("both", "uniform"):
        out["reads_drawn_uniformly"] = sub_leg("reads_drawn_uniformly", 0.0, 20260930)
    if only in ("both", "repeats"):
        out["half_of_the_reads_from_repeats"] = sub_leg("half_of_the_reads_from_repeats", args.heavy_tail_repeat_share, 20261930)
    for m_ in mps:
        m_.close()
    mps = []
    if with_e2e:
        # BASELINE config #3 verbatim on THIS genome: the product, `ngm-hip -1 a_1.fq -2 a_2.fq --affine -o out.sam`, on
        # --heavy-tail-e2e-reads reads drawn uniformly, from the first input byte to the closed SAM file (src/NGM_main.cpp:85-178 is what it
        # stands for), --e2e-runs times (median).  The input starts with the read sets of the sub-leg above, so its first records are the
        # cpu_baseline slice: the SAM of the big run is compared with the reference's SAM of that slice.
        try:
            import gc
            ne = args.heavy_tail_e2e_reads & ~1
            t0 = time.perf_counter()
            parts, have = [], 0
            for rows_ in kept_sets.get("reads_drawn_uniformly", []):
                if have < ne:
                    parts.append(rows_[:ne - have])
                    have += len(parts[-1])
            k_ = 0
            while have < ne:
                nn = min(2_000_000, ne - have) & ~1
                st_ = HL.sample_starts(G, nn // 2 if paired else nn, 400 if paired else READ_LEN, seed=20270101 + 17 * k_, repeat_share=0.0)
                parts.append(make_reads(G.contigs, nn, seed=20270102 + 17 * k_, paired=paired, subs=args.subs, indel_bases=args.indel_bases, starts=st_)[0])
                have += nn
                k_ += 1
            rows_all = np.concatenate(parts)
            del parts
            efs = [os.path.join(workdir, "he2e_1.fq"), os.path.join(workdir, "he2e_2.fq")] if paired else [os.path.join(workdir, "he2e.fq")]
            rec = write_fastq(rows_all, efs)
            del rows_all
            gc.collect()
            t_make = time.perf_counter() - t0
            esam = os.path.join(workdir, "he2e.sam")
            runs = []
            for rep_ in range(max(1, args.e2e_runs)):
                if rep_ > 0:
                    os.remove(esam)
                    os.sync()
                cmd = [B.CLI, "-r", fa] + inputs(efs) + ["-o", esam, "--workers", str(W)] + common   # (as many mapper instances as the resident leg above)
                t1 = time.perf_counter()
                rr = subprocess.run(cmd, capture_output=True, text=True)
                wall = time.perf_counter() - t1
                log = rr.stdout + rr.stderr
                if rr.returncode != 0 or "Done" not in log:
                    raise RuntimeError("ngm-hip failed: " + log[-600:])
                mio = re.search(r"Input to output: ([0-9.]+) s", log)
                mi = re.search(r"Reference and index ready: ([0-9.]+) s", log)
                mg = re.search(r"GPU kernels: ([0-9.]+) s of the ([0-9.]+) s mapping pass", log)
                runs.append({"io": float(mio.group(1)) if mio else wall, "index": float(mi.group(1)) if mi else None, "wall": wall,
                             "gpu": float(mg.group(1)) if mg else None, "pass": float(mg.group(2)) if mg else None, "log": log})
            ios = sorted(r_["io"] for r_ in runs)
            med = ios[len(ios) // 2]
            hm = [r_ for r_ in runs if r_["io"] == med][0]
            e2e = {"reads": ne, "value": ne / med, "unit": "reads/s", "seconds_first_input_byte_to_sam_closed": med,
                   "runs": {"n": len(runs), "seconds_min_median_max": [ios[0], med, ios[-1]], "reads_per_s_min_median_max": [ne / ios[-1], ne / med, ne / ios[0]],
                            "per_run": [{"input_to_output_s": r_["io"], "mapping_pass_s": r_["pass"], "gpu_kernel_s": r_["gpu"], "index_load_s": r_["index"], "process_wall_s": r_["wall"]} for r_ in runs]},
                   "index_load_s": hm["index"], "process_wall_s": hm["wall"], "sam_bytes": os.path.getsize(esam), "fastq_bytes": len(efs) * (ne // len(efs)) * rec, "make_input_s": t_make,
                   "command": " ".join(["ngm-hip"] + cmd[1:]),
                   "cli_log_tail": [l for l in hm["log"].splitlines() if "Candidate search:" in l or "Heavy-read kernel:" in l or "Candidate order replay:" in l or "Done" in l][-5:],
                   "input": "%d x %d bp %s drawn uniformly from the GRCh38-like genome, plain FASTQ, page cache warm; index from NextGenMap cache files" % (ne // 2 if paired else ne, READ_LEN, "pairs" if paired else "reads")}
            if "reads_drawn_uniformly" in kept:
                ns_, ref_body = kept["reads_drawn_uniformly"]
                same, diffs = _sam_diff(ref_body, _sam_body(esam, ns_))
                e2e["parity_vs_reference_sam"] = {"records_compared": len(ref_body), "identical_lines": same, "first_differences": diffs,
                                                  "note": "the first %d records of this run's SAM against ngm-core --affine -t N on those reads (the cpu_baseline slice of reads_drawn_uniformly)" % ns_}
            out["end_to_end"] = e2e
            for fn in efs + [esam]:
                try:
                    os.remove(fn)
                except OSError:
                    pass
        except Exception as e:
            out["end_to_end"] = {"error": str(e)[:400]}
    out["value"] = out.get("reads_drawn_uniformly", out.get("half_of_the_reads_from_repeats"))["value"]
    out["unit"] = "reads/s"
    out["note"] = "value = the sub-leg with the reads drawn uniformly; both sub-legs carry their own roofline and cpu_baseline"
    for m_ in mps:
        m_.close()
    ref.close()
    return out


def cpu_baseline_port(rows_qry, budget_s=8.0):
    import oracle_lib as O
    cores = usable_cpus()
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
    return {"value": len(idx) / ts, "unit": "scored pairs/s", "cores": cores, "cpu_quota": cores, "kind": "port",
            "sample": "%d pairs, oracle C restatement of BatchScore only, OpenMP %d threads (no candidate search)" % (len(idx), cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--genome-mbp", type=float, default=3100.0, help="synthetic genome size (GRCh38 = 3100)")
    ap.add_argument("--reads-per-step", type=int, default=1 << 20, help="reads per GPU per step")
    ap.add_argument("--cpu-sample-reads", type=int, default=2_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--personality", choices=["affine", "linear"], default="affine")
    ap.add_argument("--no-reference-rerun", action="store_true", help="skip the second -t N run of the reference program (its own run-to-run differences)")
    ap.add_argument("--heavy-tail-workers", type=int, default=4, help="mapper instances per GPU of the heavy-tailed leg (round 6, a search being one enqueue and pass 3 of the pair selection light: reads drawn uniformly 3.35 / 3.51 / 3.34 / 3.38 / 3.30 / 3.16 M reads/s with 2 / 3 / 4 / 5 / 8 / 12 instances, stress sub-leg 1.20 / 1.24 / 1.23 / 1.23 M with 4 / 6 / 8 / 12, ngm-hip on 10 M reads 2.14 / 2.47 / 2.61 / 2.19 / 2.27 M with 2 / 3 / 4 / 5 / 8; round 5 needed 8)")
    ap.add_argument("--workers", type=int, default=3, help="mapper instances (streams + host threads) per GPU (round 6, one box, 20 steps each: 2 / 3 / 4 instances 43.7 / 53.4 / 52.4 M reads/s -- a batch's host stages are 10-12 ms of its 20-24 ms, so two instances leave the GPU idle a third of the time)")
    ap.add_argument("--read-sets", type=int, default=4, help="distinct sets of reads-per-step reads the timed steps rotate through (step i maps set i mod this)")
    ap.add_argument("--read-len", type=int, default=150, help="read length (150: BASELINE config #2/#3; 250: config #5's shape)")
    ap.add_argument("--corridor", type=int, default=0, help="band width; 0: NextGenMap's 5 + 0.15 * read length")
    ap.add_argument("--layout", choices=["pe", "se"], default="pe", help="paired-end (BASELINE.json config #2) or single-end reads")
    ap.add_argument("--subs", type=float, default=0.01, help="substitution rate of the simulated reads (config #5: 0.12)")
    ap.add_argument("--indel-bases", type=float, default=0.0, help="extra share of read bases in indels (config #5: 0.03)")
    ap.add_argument("--sensitive", action="store_true", help="config #5: sensitivity 0.5 - 0.35 * 0.5 (what --sensitive does to an estimate of 0.5)")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--e2e-reads", type=int, default=10_000_000, help="reads of the end-to-end ngm-hip run (BASELINE config #3: 10 000 000)")
    ap.add_argument("--e2e-runs", type=int, default=3, help="repetitions of the end-to-end ngm-hip run (the median is reported)")
    ap.add_argument("--e2e-gz-reads", type=int, default=2_000_000, help="reads of the .fastq.gz-input and --bam-output runs of ngm-hip (0: skip)")
    ap.add_argument("--cpu-t1-reads", type=int, default=200_000, help="reads of the reference's -t 1 run (SAM cross-check)")
    ap.add_argument("--stub-mapper", action="store_true", help="CPU-only control-path run (tests)")
    ap.add_argument("--ngm-hip-exe", default=None, help="the program of the sharded end-to-end leg (default: nextgenmap_amd/ngm-hip; tests pass a stand-in)")
    ap.add_argument("--heavy-tail-mbp", type=float, default=3100.0, help="size of the GRCh38-like (heavy-tailed k-mer spectrum) genome of the second leg; 0: skip")
    ap.add_argument("--heavy-tail-steps", type=int, default=4)
    ap.add_argument("--heavy-tail-cpu-reads", type=int, default=50_000, help="reads of a heavy-tail sub-leg's cpu_baseline sample (the reference maps 350-800 of them per second; 0: none)")
    ap.add_argument("--heavy-tail-e2e-reads", type=int, default=10_000_000, help="reads of the ngm-hip run on the GRCh38-like genome, FASTQ -> closed SAM (BASELINE config #3; 0: skip)")
    ap.add_argument("--heavy-tail-repeat-share", type=float, default=0.5)
    args = ap.parse_args()
    global Q, C, READ_LEN
    READ_LEN = args.read_len
    Q = (READ_LEN | 1) + 1                                   # ReadProvider.cpp:288
    C = args.corridor if args.corridor > 0 else int(5 + 0.15 * READ_LEN)  # ReadProvider.cpp:304

    import torch
    import torch.distributed as dist
    from nextgenmap_amd import sharding
    stub = args.stub_mapper
    if not stub:
        from nextgenmap_amd.pipeline import HIT_DTYPE, Mapper, Reference
    else:
        HIT_DTYPE = np.dtype([("mapped", np.int32), ("contig", np.int32), ("pos", np.uint64), ("reverse", np.int32), ("mapq", np.int32), ("score", np.float32),
                              ("identity", np.float32), ("nm", np.int32), ("qstart", np.int32), ("qend", np.int32), ("n_candidates", np.int32),
                              ("n_best", np.int32), ("max_votes", np.float32), ("pair_flags", np.int32)])

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    pinned_cpus = 0
    if stub:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        # host threads (this one, the mapper instances', the library's pool) on the socket this rank's GPU hangs on
        from nextgenmap_amd.engine import load_library
        pinned_cpus = int(load_library().ngm_host_pin_to_device_node(local_rank))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if stub:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    def sync():
        if not stub:
            torch.cuda.synchronize()

    def barrier():
        sync()
        if world > 1:
            dist.barrier()
        sync()

    R = args.reads_per_step
    paired = args.layout == "pe"
    affine = args.personality == "affine"
    band = C + 1 if affine else C  # SeqAn's band has diagonals 0..corridor

    # ---- genome + index: rank 0 builds them once; the other ranks of the node load NextGenMap's cache files it writes ----
    wd = [tempfile.mkdtemp(prefix="ngm_bench_") if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(wd, src=0)
    workdir = wd[0]
    fa = os.path.join(workdir, "bench_ref.fa")
    gen_file = os.path.join(workdir, "genome.u8")
    t_gen = t_index = 0.0
    ref = None
    if rank == 0:
        t0 = time.perf_counter()
        contigs = make_genome(int(args.genome_mbp * 1e6), seed=20240601)
        t_gen = time.perf_counter() - t0
        t0 = time.perf_counter()
        if not stub:
            ref = Reference.from_contigs(contigs, device=local_rank, kmer=KMER, kmer_skip=2, bin_size=2)
        t_index = time.perf_counter() - t0
        if world > 1:
            np.save(os.path.join(workdir, "lens.npy"), np.array([len(c) for c in contigs], np.int64))
            with open(gen_file, "wb") as f:
                for c in contigs:
                    c.tofile(f)
            if not stub:
                with open(fa, "w") as f:
                    f.write(">stub\nACGT\n")
                ref.write_ngm_cache(fa)
    barrier()
    if rank != 0:
        lens = np.load(os.path.join(workdir, "lens.npy"))
        flat = np.memmap(gen_file, dtype=np.uint8, mode="r")
        offs = np.concatenate([[0], np.cumsum(lens)])
        contigs = [flat[offs[i]:offs[i + 1]] for i in range(len(lens))]
        t0 = time.perf_counter()
        if not stub:
            ref = Reference.from_cache(fa, device=local_rank, kmer=KMER, kmer_skip=2, bin_size=2)
        t_index = time.perf_counter() - t0

    # ---- this rank's shard of the job's reads (weak scaling: R reads per rank per step) ----------------------------------
    lo_g, hi_g = sharding.shard_range(R * world, rank, world, paired=paired)
    assert hi_g - lo_g == R
    # S distinct read sets, all resident in HBM; step i maps set i mod S (the steps do not map the same reads over and over)
    S = max(1, min(args.read_sets, args.steps if args.steps > 0 else 1))
    sets = [make_reads(contigs, R, seed=20240602 + 2 + 1000 * rank + 7919 * s_, paired=paired, subs=args.subs, indel_bases=args.indel_bases)
            for s_ in range(S)]  # (set 0: config #2's seed, one shard per rank)
    sens = 0.5 - 0.35 * 0.5 if args.sensitive else 0.5
    d_sets = [None if stub else torch.from_numpy(t_[0]).to(dev) for t_ in sets]
    last_set = (max(1, args.steps) - 1) % S
    rows, truth_c, truth_p = sets[last_set]   # what the outputs hold after the timed region (and after the isolated extra pass)
    # W mapper instances (own stream + workspace each, like NextGenMap's CS threads with their own IAlignment) work on
    # contiguous slices of the step's reads from W host threads: the host stages of one slice (pair selection, CIGAR,
    # downloads) overlap the kernels of the others
    W = max(1, min(args.workers, R // 2048))
    bounds = [(R * w // W) & ~1 for w in range(W)] + [R]
    out = (np.zeros(R, HIT_DTYPE), np.zeros((R, 4 * Q), np.uint8), np.zeros((R, 4 * Q), np.uint8))

    class StubMapper:
        """control-path stand-in (tests): every read 'maps' where the simulator put it, 1 read in 1000 does not"""
        def map(self, rw, ow, lo):
            h = ow[0]
            h["mapped"] = 1; h["contig"] = truth_c[lo:lo + len(h)]; h["pos"] = truth_p[lo:lo + len(h)]; h["mapq"] = 60; h["pair_flags"] = 1
            h["mapped"][(np.arange(lo, lo + len(h)) + lo_g) % 1000 == 999] = 0
        def last_kernel_ms(self): return [1.0] * 8
        def cs_counters(self): return [138 * R // W, 4000 * R // W, R // W]
        def close(self): pass

    mps, views = [], []
    for w in range(W):
        lo, hi = bounds[w], bounds[w + 1]
        if stub:
            mps.append(StubMapper())
            views.append(([t_[0][lo:hi] for t_ in sets], None, tuple(o[lo:hi] for o in out), lo))
            continue
        kw = dict(gap_read=33, gap_ref=33, gap_extend=3, personality=1) if affine else {}
        mps.append(Mapper(ref, Q, C, sensitivity=sens, **kw))
        views.append(([t_[0][lo:hi] for t_ in sets], [d_[lo:hi] for d_ in d_sets], tuple(o[lo:hi] for o in out), lo))

    def worker(w, steps, acc, first=0):
        rws, dws, ow, lo = views[w]
        k = np.zeros(8)
        for i_ in range(steps):
            s_ = (first + i_) % S
            if stub:
                mps[w].map(rws[s_], ow, lo)
            elif paired:
                mps[w].map_pe_raw(rws[s_], dws[s_], ow)
            else:
                mps[w].map_se_raw(rws[s_], dws[s_], ow)
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

    # set-up, not warm-up: every read set once, so that the mappers' device and page-locked buffers have seen their largest batch
    # (they grow to what a batch needs; with fewer warm-up steps than read sets that growth -- hipFree + hipMalloc, device-wide
    # synchronisations -- would fall into the timed steps)
    if S > 1:
        run(S)
    run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    kms = run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    kms /= max(1, args.steps)  # GPU time per step (R reads), summed over the W streams' launches
    hits = out[0]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # one extra, untimed pass of instance 0 alone: kernel durations without the other streams' kernels sharing the GPU
    worker(0, 1, iso := [None] * W, first=last_set)
    iso_ms = iso[0]
    iso_ctr = mps[0].cs_counters()
    ctr = np.sum([m_.cs_counters() for m_ in mps], axis=0)
    kmers, hits_voted, n_cand = int(ctr[0]), int(ctr[1]), int(ctr[2])
    mapped = hits["mapped"] == 1
    correct = mapped & (hits["contig"] == truth_c) & (np.abs(hits["pos"].astype(np.int64) - truth_p) <= C // 2 + 4)
    # ONE collective of the path -- the int64[8] mapping statistics, summed over the ranks (RCCL over xGMI)
    local = {"reads": R, "mapped": int(mapped.sum()), "unmapped": int((~mapped).sum()), "written": R}
    if paired:
        sel = ((hits["pair_flags"][0::2] & 1) != 0) & mapped[0::2] & mapped[1::2]
        ins = np.abs(hits["pos"][0::2].astype(np.int64) - hits["pos"][1::2].astype(np.int64)) + READ_LEN
        local.update(pairs_total=R // 2, pairs_broken=int(((hits["pair_flags"][0::2] & 2) != 0).sum()), insert_sum=int(ins[sel].sum()), insert_cnt=int(sel.sum()))
    stats = sharding.reduce_stats(local, device=dev)
    # the collective saw every rank: the summed read count is R per rank
    assert stats["reads"] == R * world, "stats all-reduce: %r reads, expected %d x %d" % (stats["reads"], R, world)
    stats["ranks_seen"] = stats["reads"] // R

    if rank == 0:
        value = R * world * args.steps / elapsed
        score_cells = n_cand * READ_LEN * band
        align_cells = int(mapped.sum()) * READ_LEN * band
        b_cs = 20 * kmers + 4 * hits_voted + 16 * n_cand
        n_aln = int(mapped.sum())
        # the dominant kernel of THIS workload (largest GPU time per step) and its algorithmic bytes
        kernels = {"candidate_search": (kms[0], b_cs, "cs_canon_kernel / cs_fast2_kernel (candidate search): 20 B/k-mer + 4 B/index hit + 16 B/candidate"),
                   "sw_score": (kms[2], n_cand * (Q + Q + C + 4), "sw_*score*_kernel (BatchScore): B_score = q + (q + c) + 4 bytes per pair"),
                   "sw_align": (kms[5] + kms[6], n_aln * (Q + Q + C + 8 + 4 * (2 * Q + C + 1)), "sw_*align*_kernel + traceback (BatchAlign): B_align = q + (q + c) + 8 + 4 (2q + c + 1) bytes per pair")}
        dom = max(kernels, key=lambda k_: kernels[k_][0])
        dom_ms, dom_bytes, dom_note = kernels[dom]
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # HBM bytes per launch of the candidate search: NOT measured in this run -- taken from the committed PMC passes
        # (profiles/*_pmc_traffic.json, keyed by kernel name and grid size; collected with this default workload)
        # (only the pass of THIS round's kernel counts: the file of the round, the exact template instance the default workload launches --
        # anything else and the field is null rather than another kernel's figure)
        traffic = traffic_source = None
        if dom == "candidate_search" and int(args.genome_mbp) == 3100 and READ_LEN == 150 and not stub and paired and affine:
            fn = PMC_TRAFFIC_FILE
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", fn)))
            except Exception:
                tj = {}
            for k, v in tj.items():
                if k.startswith(CS_DEFAULT_KERNEL + "|"):
                    traffic, traffic_source = v, "profiles/" + fn + " [" + k + "]"
        line = {
            "metric": "mapped reads/sec + SW Gcells/sec, 150bp vs GRCh38, at 1/2/4/8 MI355X",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": ("%d x %dbp %s synthetic reads per GPU per step (%.1f %% substitutions%s) vs a synthetic %.0f Mbp genome (24 contigs, repeat "
                                    "families, N runs; GRCh38 itself is not available offline): candidate search (k=13, skip 2, s=%.3f) + "
                                    "score + %s/MAPQ + align with traceback + CIGAR; reads resident in HBM; the steps rotate through %d distinct read sets")
                       % (R, READ_LEN, "PE (insert ~N(350,35), FR)" if paired else "SE", 100 * args.subs,
                          ", %.0f %% indel bases" % (100 * args.indel_bases) if args.indel_bases else "", args.genome_mbp, sens,
                          "pair selection (top1PE)" if paired else "top-1", S),
                       "qry_max_len": Q, "corridor": C, "scoring": "affine (SeqAn banded Gotoh) 10/15/33/3 local" if affine else "linear 10/15/20/20 local", "reads_per_step_per_gpu": R, "mapper_instances_per_gpu": W, "host_cpus_per_rank_numa_pinned": pinned_cpus,
                       "parallelism": "reads sharded x%d (nextgenmap_amd.sharding.shard_range), genome+index replicated per GPU (built by rank 0, loaded from NextGenMap cache files by the others)" % world},
            "sw_gcells_per_s": {"score_kernel": score_cells / (kms[2] * 1e-3) / 1e9 if kms[2] > 0 else None,
                                "align_kernel": align_cells / (kms[5] * 1e-3) / 1e9 if kms[5] > 0 else None},
            "kernel_ms": {"candidate_search": kms[0], "gather_score": kms[1], "sw_score": kms[2], "select": kms[3], "gather_align": kms[4],
                          "sw_align": kms[5], "traceback": kms[6], "all_kernels": float(kms[:7].sum()),
                          "candidate_search_stage_incl_host_sync": kms[7]},
            "per_read": {"candidates": n_cand / R, "index_hits": hits_voted / R, "kmers": kmers / R},
            "accuracy_rank0_shard": {"mapped": float(mapped.mean()), "within_band_of_truth": float(correct.mean()), "mapq_gt0": float((hits["mapq"] > 0).mean())},
            "setup_s": {"genome_generation": t_gen, "encode+index_build": t_index, "index_entries": ref.index_entries if ref else 0},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source, "kernel": dom, "kernel_note": dom_note, "kernel_ms_per_step": float(dom_ms),
                         "candidate_search": {"achieved": b_cs / (kms[0] * 1e-3) / 1e9 if kms[0] > 0 else 0.0, "frac": (b_cs / (kms[0] * 1e-3) / 1e9 / HBM_PEAK_GBS) if kms[0] > 0 else 0.0,
                                              "bytes_per_launch": b_cs / W, "launches_per_step": W, "ms_per_launch": float(kms[0] / W),
                                              "isolated": {"achieved": (20 * iso_ctr[0] + 4 * iso_ctr[1] + 16 * iso_ctr[2]) / (iso_ms[0] * 1e-3) / 1e9 if iso_ms[0] > 0 else 0.0, "ms": float(iso_ms[0]),
                                                           "reads": int(bounds[1] - bounds[0]),
                                                           "note": "one launch of one mapper instance with nothing else on the GPU (untimed extra pass)"}},
                         "note": "`kernel` = the kernel with the largest GPU time per step of this workload; achieved = its algorithmic bytes (SURVEY.md 8d) / its "
                                 "HIP-event time on the launch stream.  Candidate search gathers from the canonical pair buckets (one 128-byte line per k-mer pair + "
                                 "16-byte chunks beyond 31 positions); random gathers on MI355X are bound by ~50 G requests/s "
                                 "(profiles/r02_gather_calibration.txt) -- and the kernel itself issues VALU instructions 77-85 % of the time "
                                 "(profiles/r03_sq_counters_cs_and_dp_kernels.txt, profiles/r03_valu_lds_issue_rate_calibration.txt: 4.3 cycles per wave "
                                 "instruction).  `traffic` = FETCH_SIZE + WRITE_SIZE of the committed rocprofv3 PMC pass named in traffic_source -- not "
                                 "measured in this run; FETCH_SIZE tallies a request as 64 bytes, the 128-byte first lines too, so the bytes moved are "
                                 "up to 64 B x k-mers per read more.  The SW kernels are VALU-bound, see sw_gcells_per_s"},
            # SURVEY.md 8d's whole-path figure: (pairs * B_score + alignments * B_align + B_cs) per second of wall time
            "path_algorithmic_gbs": (n_cand * (Q + Q + C + 4) + int(mapped.sum()) * (Q + Q + C + 8 + 4 * (2 * Q + C + 1)) + b_cs) * world
                                    / (elapsed / args.steps) / 1e9,
            "stats_allreduce": stats,
        }
        if world > 1 or stub:
            line["cpu_baseline"] = None  # the host baseline is timed on rank 0 of a 1-GPU run only
            line["end_to_end"] = None
            if world > 1 and not args.no_end_to_end and (not stub or args.ngm_hip_exe):
                # the real config 4: one 10 M-read input through the product, one process per GPU (the other ranks wait at the barrier below)
                try:
                    if args.ngm_hip_exe:
                        exe = args.ngm_hip_exe
                    else:
                        from nextgenmap_amd import build as B_
                        exe = B_.CLI
                    line["end_to_end"] = sharded_end_to_end(workdir, contigs, args, paired, affine, sens, world, exe)
                except Exception as e:
                    line["end_to_end"] = {"error": str(e)[:400]}
        else:
            e2e_base = None
            if not args.no_end_to_end:
                try:
                    line["end_to_end"], e2e_base = end_to_end(ref, contigs, workdir, args, paired, affine, sens)
                except Exception as e:
                    line["end_to_end"] = {"error": str(e)[:400]}
            if e2e_base is not None:
                line["cpu_baseline"] = e2e_base
            elif not args.no_cpu_baseline:
                # no reference program (not built, or the linear personality, which needs an OpenCL CPU device it does not have here):
                # the oracle's C restatement of the score stage only
                line["cpu_baseline"] = cpu_baseline_port(rows[:8192])
                line["cpu_baseline"]["note"] = ("reference program not run: " + ("ngm-core only runs --affine on this host" if not affine else
                                                "oracle/_ref/ngm/ngm-core not built or --no-end-to-end given"))
    for m_ in mps:
        m_.close()
    if ref is not None:
        ref.close()
    if rank == 0:
        if world == 1 and not stub and args.heavy_tail_mbp > 0 and READ_LEN == 150:
            del contigs, rows, sets, d_sets, views
            try:
                line["heavy_tailed_genome"] = heavy_tail_leg(args, dev, local_rank, paired, affine, sens, workdir)
            except Exception as e:
                line["heavy_tailed_genome"] = {"error": str(e)[:400]}
        print(json.dumps(line))
    barrier()
    if rank == 0:
        import shutil
        shutil.rmtree(workdir, ignore_errors=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
