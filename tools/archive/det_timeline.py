#!/usr/bin/env python3
"""Prints the kernel timeline of one deterministic-mode step out of a rocprofv3 kernel trace
(csv) of tools/archive/exp_det.py: start offset, duration, grid, name.  usage: det_timeline.py TRACE.csv [STEP]"""
import csv
import re
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
heads = [i for i, r in enumerate(rows) if "k_det_phase1" in r["Kernel_Name"] and int(r["Grid_Size_X"]) >= 1 << 20]
at = int(sys.argv[2]) if len(sys.argv) > 2 else len(heads) // 2
seg = rows[heads[at]:heads[at + 1]]
t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.2f} +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.2f} us  "
          f"grid {r['Grid_Size_X']:>8}  {name[:60]}")
print(f"step: {(int(rows[heads[at + 1]]['Start_Timestamp']) - t0) / 1e3:.2f} us")
