#!/usr/bin/env python3
"""Turn rocprofv3 (ROCm 7.2, rocpd sqlite output) runs of bench.py into the small text summaries that
are committed under profiles/.

  python profiles/summarize_rocprof.py <tag> <stats_db> [<fetch_db> <write_db>]

stats_db : rocprofv3 --kernel-trace --stats           -> <tag>_kernel_stats.csv
fetch_db : rocprofv3 --pmc FETCH_SIZE --kernel-trace  \
write_db : rocprofv3 --pmc WRITE_SIZE --kernel-trace  /-> <tag>_pmc_hbm.csv + <tag>_pmc_traffic.json
HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB, and on
gfx950 FETCH_SIZE counts a coalesced streaming read at half its bytes (128-B requests tallied as 64 B),
so read bytes = 2 * FETCH_SIZE * 1024 (confirmed here: see the "expected" column notes in DESIGN.md).
Round 2 calibration (profiles/r02_gather_calibration.txt, profiles/tools/gather_calib.hip): FETCH_SIZE = memory requests x 64 B;
a random gather of <= 64 bytes is ONE 64-byte request (FETCH_SIZE exact), an aligned 128-byte gather one 128-byte request
(FETCH_SIZE half).  The candidate-search kernels gather 8-byte headers and 32-byte list segments: factor 1 for them
(kernel names containing "cs_"), factor 2 for the streaming kernels.
"""


def fetch_factor(name):
    return 1 if "cs_" in name else 2
import csv
import json
import os
import sqlite3
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def short(name):
    n = name.split("(")[0].replace("void ", "")
    return n if len(n) < 90 else n[:87] + "..."


def main():
    tag, stats_db = sys.argv[1], sys.argv[2]
    if stats_db != "-":   # ("-": only the PMC passes)
        cur = sqlite3.connect(stats_db).cursor()
        rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
        with open(os.path.join(HERE, tag + "_kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
            for name, calls, tot, avg, pct in rows:
                w.writerow([short(name), calls, "%.3f" % tot, "%.3f" % avg, "%.2f" % pct])
    if len(sys.argv) >= 5:
        agg = {}
        for db, ctr in ((sys.argv[3], "FETCH_SIZE"), (sys.argv[4], "WRITE_SIZE")):
            c = sqlite3.connect(db).cursor()
            q = ("select kernel_name, grid_size, count(*), avg(value), avg(duration) from counters_collection "
                 "where counter_name = ? group by kernel_name, grid_size")
            for name, grid, n, val, dur in c.execute(q, (ctr,)):
                if "ngm::" not in name:
                    continue
                agg.setdefault((short(name), grid), {})[ctr] = (n, val, dur)
        traffic = {}
        with open(os.path.join(HERE, tag + "_pmc_hbm.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "grid_size", "dispatches", "FETCH_SIZE_KiB_avg", "WRITE_SIZE_KiB_avg",
                        "hbm_read_bytes(FETCH x calibrated factor: 1 gathers, 2 streaming)", "hbm_write_bytes", "hbm_bytes_per_launch"])
            for (name, grid), d in sorted(agg.items()):
                fe = d.get("FETCH_SIZE", (0, 0.0, 0))[1]
                wr = d.get("WRITE_SIZE", (0, 0.0, 0))[1]
                rb, wb = fetch_factor(name) * fe * 1024, wr * 1024
                w.writerow([name, grid, d.get("FETCH_SIZE", (0,))[0], "%.1f" % fe, "%.1f" % wr, int(rb), int(wb), int(rb + wb)])
                traffic["%s|grid=%d" % (name, grid)] = int(rb + wb)
        with open(os.path.join(HERE, tag + "_pmc_traffic.json"), "w") as f:
            json.dump(traffic, f, indent=1, sort_keys=True)
        # round 6: the candidate-search GROUP per batch (every kernel of one ngm_mapper search: fast path, queue kernels, heavy classes, exact
        # kernels, compaction), and the order replay beside it -- bench.py's heavy-tailed leg prices the group's algorithmic bytes against this
        batches = sum(d.get("FETCH_SIZE", (0,))[0] for (name, grid), d in agg.items() if "compact_candidates_kernel" in name)
        if batches:
            group, replay = {}, {}
            for (name, grid), d in agg.items():
                fe, wr = d.get("FETCH_SIZE", (0, 0.0, 0)), d.get("WRITE_SIZE", (0, 0.0, 0))
                total = fetch_factor(name) * fe[1] * 1024 * fe[0] + wr[1] * 1024 * wr[0]
                if "cs_order" in name:
                    replay[name] = replay.get(name, 0) + total
                elif "cs_" in name or "compact_candidates" in name:
                    group[name] = group.get(name, 0) + total
            with open(os.path.join(HERE, tag + "_cs_traffic.json"), "w") as f:
                json.dump({"batches": batches, "candidate_search_bytes_per_batch": int(sum(group.values()) / batches),
                           "order_replay_bytes_per_batch": int(sum(replay.values()) / batches),
                           "candidate_search_by_kernel_per_batch": {k: int(v / batches) for k, v in sorted(group.items())},
                           "order_replay_by_kernel_per_batch": {k: int(v / batches) for k, v in sorted(replay.items())},
                           "note": "FETCH_SIZE x calibrated factor + WRITE_SIZE, summed over the dispatches of the pass and divided by the "
                                   "number of searches (dispatches of compact_candidates_kernel)"}, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
