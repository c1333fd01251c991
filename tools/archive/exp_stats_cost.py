#!/usr/bin/env python3
"""Measurement tool: what do the per-block statistics atomics (one global atomic per block and statistic,
all blocks on the same few addresses) cost the kernels?  Fused rollout in 20-step launches and the
deterministic step, with the statistics vectors passed and with NULL."""
import importlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("2048_q-learning_amd")
N = pkg._native
L = N.lib()
dev = torch.device("cuda:0")
B = 1 << 20
for mode in ("fused", "det"):
    for with_stats in (True, False, True, False):
        env = pkg.BatchedGame2048Env(B, seed=0, device=dev)
        agent = pkg.BatchedQLearningAgent(1000, learning_rate=0.1, discount_factor=0.99, exploration_rate=0.95,
                                          capacity_log2=30, seed=0, device=dev)
        agent.fused_rollout(env, 256, play_only=True)
        si, sf = agent.stats_i, agent.stats_f
        go = agent.fused_rollout if mode == "fused" else agent.deterministic_rollout
        go(env, 20)
        if not with_stats:
            agent.stats_i = agent.stats_f = None
        launches = 8 if mode == "fused" else 3
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            go(env, 20)
        e1.record(); torch.cuda.synchronize()
        agent.stats_i, agent.stats_f = si, sf
        print(json.dumps({"mode": mode, "statistics": with_stats, "steps_per_launch": 20,
                          "us_per_step": round(e0.elapsed_time(e1) * 1e3 / (20 * launches), 2)}), flush=True)
        del agent, env
        torch.cuda.empty_cache()
