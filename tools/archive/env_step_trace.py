#!/usr/bin/env python3
"""Per-dispatch durations of the env-step kernels out of a rocprofv3 kernel trace (csv) of
tools/archive/exp_env_step.py: median and mean by (kernel, grid).  usage: env_step_trace.py TRACE.csv"""
import collections
import csv
import re
import statistics
import sys

by = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_env_step" not in r["Kernel_Name"]:
        continue
    name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
    by[(name, int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, grid), d in sorted(by.items(), key=lambda kv: (kv[0][0], kv[0][1])):
    d = d[len(d) // 5:]
    print(f"{name:34s} grid {grid:>9}  n {len(d):4d}  median {statistics.median(d):7.2f} us  mean {statistics.fmean(d):7.2f} us")
