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
            short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-40:]
            acc[(short, r.get("Counter_Name"))].append((int(r.get("Dispatch_Id", 0)), float(r.get("Counter_Value", 0))))
    print("==", os.path.basename(d))
    for (k, c), v in sorted(acc.items()):
        if not k.startswith("k_"): continue
        per = collections.defaultdict(float)
        for did, val in v: per[did] += val
        vals = [per[k2] for k2 in sorted(per)]
        tail = vals[len(vals)//2:]            # timed half (after warm-up)
        print(f"  {k:32s} {c:24s} n={len(vals):3d} mean_all={sum(vals)/len(vals):.4e} mean_tail={sum(tail)/len(tail):.4e}")

# traffic per env-step of the fused kernel (timed launches), for bench.py's roofline.traffic
import json
def _mean(d, kernel, counter):
    acc = collections.defaultdict(float)
    for f in glob.glob(os.path.join(root, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
                acc[int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    vals = [acc[k] for k in sorted(acc)]
    vals = vals[1:]                       # drop the warm-up launch
    return (sum(vals) / len(vals), int(list(csv.DictReader(open(f)))[0]["Grid_Size"])) if vals else (None, 0)
try:
    fetch, _ = _mean("pass1", "k_fused_rollout", "FETCH_SIZE")
    write, _ = _mean("pass2", "k_fused_rollout", "WRITE_SIZE")
    if fetch is not None and write is not None and len(sys.argv) > 3:
        boards, steps_per_launch = int(sys.argv[2]), int(sys.argv[3])
        per_launch = (fetch + write) * 1024.0 + 16.0 * boards  # + the half of the 32 B/lane wide loads FETCH_SIZE misses on gfx950
        out = {"bytes_per_env_step": per_launch / (boards * steps_per_launch),
               "fetch_kb_per_launch": fetch, "write_kb_per_launch": write, "boards": boards,
               "steps_per_launch": steps_per_launch,
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `bench.py --cpu-seconds 0`; "
                         "(FETCH_SIZE + WRITE_SIZE) * 1024 + 16 B/board for the wide board+aux loads that gfx950 "
                         "counts at half"}
        print("TRAFFIC_JSON " + json.dumps(out))
except Exception as e:  # summary only
    print("traffic: n/a", e)
