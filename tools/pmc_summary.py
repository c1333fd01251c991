#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection CSVs: per kernel, per counter: mean over dispatches."""
import csv, glob, os, sys, collections
root = sys.argv[1]
for d in sorted(glob.glob(os.path.join(root, "pass*"))):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            short = name.replace("(anonymous namespace)::", "").split("(")[0][-40:]
            acc[(short, r.get("Counter_Name"))].append((int(r.get("Dispatch_Id", 0)), float(r.get("Counter_Value", 0))))
    print("==", os.path.basename(d))
    for (k, c), v in sorted(acc.items()):
        if not k.startswith("k_"): continue
        per = collections.defaultdict(float)
        for did, val in v: per[did] += val
        vals = [per[k2] for k2 in sorted(per)]
        tail = vals[len(vals)//2:]            # timed half (after warm-up)
        print(f"  {k:32s} {c:24s} n={len(vals):3d} mean_all={sum(vals)/len(vals):.4e} mean_tail={sum(tail)/len(tail):.4e}")
