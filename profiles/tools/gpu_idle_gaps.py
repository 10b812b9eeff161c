#!/usr/bin/env python3
"""GPU idle time by cause: reads a rocprofv3 --kernel-trace CSV, merges the kernels of all streams into busy intervals and lists
the idle gaps by (kernel before the gap -> kernel after it).  python profiles/tools/gpu_idle_gaps.py <kernel_trace.csv> [skip_first_ms]"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
rows.sort()
t0 = rows[0][0]
skip = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0
rows = [r for r in rows if r[0] - t0 >= skip]
first = next((i for i, r in enumerate(rows) if "cs_canon_kernel" in r[2]), 0)   # the mapping loop: from its first candidate search on
rows = rows[first:]
busy = 0
gaps = collections.Counter(); cnt = collections.Counter()
cur_end, cur_name = rows[0][1], rows[0][2]
start_all = rows[0][0]
for s, e, n in rows[1:]:
    if s > cur_end:
        if s - cur_end < 50e6:   # (longer: between phases of the program)
            gaps[(cur_name, n)] += s - cur_end; cnt[(cur_name, n)] += 1
    if e > cur_end:
        cur_end, cur_name = e, n
for s, e, n in rows:
    busy += e - s
span = cur_end - start_all
tot_gap = sum(gaps.values())
print("span %.1f ms, kernel time (summed, overlaps counted twice) %.1f ms, idle in gaps < 50 ms: %.1f ms (%.1f %% of the span)" % (span / 1e6, busy / 1e6, tot_gap / 1e6, 100.0 * tot_gap / span))
for (a, b), g in gaps.most_common(25):
    print("%8.2f ms  %6d x %7.1f us   %s  ->  %s" % (g / 1e6, cnt[(a, b)], g / cnt[(a, b)] / 1e3, a, b))
