#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection CSVs: per kernel and counter, the mean over the timed
dispatches (the LAST `timed` dispatches of the kernel: the bench's timed regions come last), per
dispatch and per env-step.  Usage: pmc_summary.py <pmc dir> <bench json of one pass>"""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
bench = json.load(open(sys.argv[2])) if len(sys.argv) > 2 and os.path.getsize(sys.argv[2]) else None
timed = steps_per_dispatch = None
if bench:
    cfg = bench["config"]
    launches = bench["roofline"]["launches"]
    timed = launches * cfg["repeats"]
    steps_per_dispatch = cfg["boards_per_gpu"] * bench["steps"] / launches
summary = {}
for d in sorted(glob.glob(os.path.join(root, "pass*"))):
    if not os.path.isdir(d):
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-44:]
            acc[(short, r.get("Counter_Name"))][int(r.get("Dispatch_Id", 0))] += float(r.get("Counter_Value", 0))
    print("==", os.path.basename(d))
    for (k, c), per in sorted(acc.items()):
        if "k_fused_rollout" not in k and "k_table" not in k:
            continue
        vals = [per[i] for i in sorted(per)]
        tail = vals[-timed:] if (timed and "k_fused_rollout" in k) else vals
        mean = sum(tail) / len(tail)
        line = f"  {k:40s} {c:36s} n={len(vals):3d} mean_timed={mean:.4e}"
        if steps_per_dispatch and "k_fused_rollout" in k:
            line += f"  per_env_step={mean / steps_per_dispatch:.4f}"
            summary[c] = mean / steps_per_dispatch
        print(line)
if bench and "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
    cfg = bench["config"]
    S = cfg["steps_per_launch"]
    # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 counts wide coalesced streaming reads at half:
    # the 32 B/lane/launch of board + aux loads -> + 16 B per board per launch
    per_step = (summary["FETCH_SIZE"] + summary["WRITE_SIZE"]) * 1024.0 + 16.0 / S
    out = {"bytes_per_env_step": per_step,
           "fetch_bytes_per_env_step": summary["FETCH_SIZE"] * 1024.0,
           "write_bytes_per_env_step": summary["WRITE_SIZE"] * 1024.0,
           "requests_per_env_step": {k: v for k, v in summary.items() if k.startswith("TCC_")},
           "config": {"boards": cfg["boards_per_gpu"], "steps_per_launch": S,
                      "cap_log2": cfg["table_capacity_log2"],
                      "board_size": 4 if "4x4" in cfg["workload"] else 5, "eps": cfg["epsilon"],
                      "strict_td": cfg["td_write"] != "store (last writer wins)"},
           "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `"
                     + open(os.path.join(root, "command.txt")).read().strip() + "`, timed dispatches only; "
                     "(FETCH_SIZE + WRITE_SIZE) * 1024 + 16 B/board/launch for the wide board+aux loads "
                     "that gfx950 counts at half"}
    print("TRAFFIC_JSON " + json.dumps(out))
