#!/usr/bin/env python3
"""Round 6, one A/B in one process: do bench.py's companions run slower because they start right
after a 32 GiB chunked table was released -- the driver wipes released memory at ~40 GB/s, 0.8 s for
32 GiB, and a companion's timed regions are 1 ms each -- or is the 5x5 companion's drift on the driver's boxes
(0.329 / 0.324 / 0.316 in rounds 3-5) box variance?

Hypothesis (written down before the run): if the wipe runs under the companion's regions, (ii) is slower than (i)
and (iii) by more than the spread between (i) and (iii).

  (i)    the 5x5 companion FIRST in the process: nothing has been freed yet
  --     the 4x4 main line (maps a 32 GiB table in chunks, measures, releases it)
  (ii)   the 5x5 companion right after that release (what bench.py does)
  (iii)  the same after a 2 s sleep (32 GiB are wiped in 0.8 s)
  (iv)   the same after another 10 s
    python tools/exp_wipe_ab.py > profiles/r06_wipe_ab.jsonl"""
import argparse
import importlib
import importlib.util
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
pkg = importlib.import_module("2048_q-learning_amd")
spec = importlib.util.spec_from_file_location("bench", os.path.join(REPO, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

args = bench.parse_args(["--steps", "20", "--warmup", "5"])
args.prep_steps = max(64, args.prep_steps)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
shard = pkg.weak_shard(args.boards_per_gpu, 1, 0)
reducer = pkg.StatsAllReduce(dev)
cap = 30


def run(name, n, sleep_before=0.0):
    if sleep_before:
        time.sleep(sleep_before)
    free0, _ = torch.cuda.mem_get_info(dev)
    t0 = time.perf_counter()
    m = bench.measure(pkg, torch, args, dev, shard, 1, eps=args.eps, cap_log2=cap, placement="auto", steps=20, warmup=5,
                      repeats=3 if n == 5 else 5, S=20, reducer=reducer, board_size=n)
    s = bench.summarise(m, shard, 20, bench.ALGO_BYTES_FUSED_4X4 if n == 4 else bench.ALGO_BYTES_FUSED_5X5)
    print(json.dumps({"case": name, "board_size": n, "sleep_before_s": sleep_before, "free_GiB_before": round(free0 / 2**30, 1),
                      "wall_s": round(time.perf_counter() - t0, 2), "ms_per_step": s["ms_per_step"], "region_ms": s["region_ms"],
                      "avg_launch_ms": s["avg_launch_s"] * 1e3, "roofline_frac": s["achieved_gbs"] / bench.HBM_PEAK_GBS,
                      "placement": m["placement"]}), flush=True)


run("(i) 5x5 first, nothing freed yet", 5)
run("main 4x4 line", 4)
run("(ii) 5x5 right after the 32 GiB table was released", 5)
run("(iii) 5x5 after a 2 s sleep", 5, 2.0)
run("(iv) 5x5 after another 10 s", 5, 10.0)
