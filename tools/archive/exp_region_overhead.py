#!/usr/bin/env python3
"""Measurement tool: what the timed region of bench.py holds besides its launches -- one 20-step
launch bracketed as bench.py brackets it, with and without the statistics reduction (N = 1)."""
import importlib
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("2048_q-learning_amd")
dev = torch.device("cuda:0")
B, K = 1 << 20, 20
env = pkg.BatchedGame2048Env(B, seed=0, device=dev)
agent = pkg.BatchedQLearningAgent(1000, exploration_rate=0.95, capacity_log2=30, seed=0, device=dev)
agent.fused_rollout(env, 256, play_only=True)
agent.ctr = env.ctr
agent.fused_rollout(env, 5)
reducer = pkg.StatsAllReduce(dev)
reducer.start(agent.stats_i, agent.stats_f); reducer.wait()
for mode in ("launch only", "launch + reduce", "launch only", "launch + reduce"):
    walls, kernels = [], []
    for _ in range(7):
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        agent.fused_rollout(env, K)
        e1.record()
        if mode == "launch + reduce":
            reducer.start(agent.stats_i, agent.stats_f)
            reducer.wait()
        torch.cuda.synchronize(dev)
        walls.append((time.perf_counter() - t0) * 1e6)
        kernels.append(e0.elapsed_time(e1) * 1e3)
    walls.sort(); kernels.sort()
    print(json.dumps({"mode": mode, "wall_us_median": round(walls[3], 1), "events_us_median": round(kernels[3], 1),
                      "outside_the_launch_us": round(walls[3] - kernels[3], 1)}), flush=True)
