#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection CSVs: per kernel and counter, the mean over the timed
dispatches (the LAST `timed` dispatches of the kernel: the bench's timed regions come last), per
dispatch and per env-step.  Usage: pmc_summary.py <pmc dir> <bench json of one pass>"""
import collections
import csv
import glob
import json
import os
import re
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
        m = re.search(r"k_fused_rollout<\d, (\d)", k)
        if m and int(m.group(1)) & 4:      # the learner-less instantiation (input synthesis): not the measured kernel
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
    # Read side: FETCH_SIZE (KiB) tallies every fabric read request at 64 B, but the requests are
    # 128 B (TCC_EA0_RDREQ_128B: every L2 miss of this kernel, the random 16-B probes included --
    # MI355X_MICROARCH.md "HBM": FETCH_SIZE reports exactly half on such reads).  When the pass with
    # the request-size counters is there the bytes come from it, otherwise FETCH_SIZE is doubled.
    # Write side: WRITE_SIZE is exact (32-B write-backs + 64-B atomics; cross-checked below).
    rd = {k: summary.get(f"TCC_EA0_RDREQ_{k}_sum") for k in ("32B", "64B", "128B")}
    if all(v is not None for v in rd.values()):
        read_bytes = 32.0 * rd["32B"] + 64.0 * rd["64B"] + 128.0 * rd["128B"]
        read_how = "32 x RDREQ_32B + 64 x RDREQ_64B + 128 x RDREQ_128B"
    else:
        read_bytes = 2.0 * summary["FETCH_SIZE"] * 1024.0
        read_how = "2 x FETCH_SIZE (128-B requests tallied at 64 B)"
    write_bytes = summary["WRITE_SIZE"] * 1024.0
    import hashlib
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "2048_q-learning_amd", "csrc")
    sha = hashlib.sha256()
    for name in ("q2048_kernels.hip", "q2048_core.hpp", "q2048_core5.hpp", "q2048_luts.inc"):   # = bench.KERNEL_SOURCES
        sha.update(open(os.path.join(csrc, name), "rb").read())
    out = {"kernel_sources_sha16": sha.hexdigest()[:16],        # bench.py reports this traffic only for these sources
           "bytes_per_env_step": read_bytes + write_bytes,
           "read_bytes_per_env_step": read_bytes, "write_bytes_per_env_step": write_bytes,
           "fetch_size_bytes_per_env_step_uncorrected": summary["FETCH_SIZE"] * 1024.0,
           "requests_per_env_step": {k: v for k, v in summary.items() if k.startswith("TCC_")},
           "config": {"boards": cfg["boards_per_gpu"], "steps_per_launch": S,
                      "cap_log2": cfg["table_capacity_log2"],
                      "board_size": 4 if "4x4" in cfg["workload"] else 5, "eps": cfg["epsilon"],
                      "strict_td": cfg["td_write"] != "store (last writer wins)",
                      **({"prefill_load": cfg["prefill_load"]} if cfg.get("prefill_load") else {})},
           "source": "rocprofv3 --kernel-trace --pmc, one counter set per pass, on `"
                     + open(os.path.join(root, "command.txt")).read().strip() + "`, timed dispatches only; "
                     "read bytes = " + read_how + ", write bytes = WRITE_SIZE x 1024"}
    if "TCC_EA0_WRREQ_64B_sum" in summary and "TCC_EA0_WRREQ_sum" in summary:
        out["write_bytes_cross_check"] = 64.0 * summary["TCC_EA0_WRREQ_64B_sum"] + 32.0 * (
            summary["TCC_EA0_WRREQ_sum"] - summary["TCC_EA0_WRREQ_64B_sum"])
    print("TRAFFIC_JSON " + json.dumps(out))
