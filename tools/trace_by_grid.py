#!/usr/bin/env python3
"""Per (kernel, grid size) median / mean / total duration out of a rocprofv3 kernel trace CSV.
Usage: trace_by_grid.py <kernel_trace.csv> [out.txt]"""
import collections
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-48:]
    grid = int(r["Grid_Size"]) if "Grid_Size" in r else int(r.get("Grid_Size_X", 0))
    acc[(name, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else None
for (n, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    line = (f"{n:50s} grid {g:10d} n={len(v):5d} median_us={statistics.median(v):9.2f} "
            f"mean_us={sum(v) / len(v):9.2f} total_ms={sum(v) / 1e3:9.2f}")
    print(line)
    if out:
        out.write(line + "\n")
