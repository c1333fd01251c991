#!/usr/bin/env python3
"""Measurement tool: us per 1 Mi-board step of the fused rollout, launch by launch (20 steps each), as a table
fills: 2^28 slots from load 0 to ~0.5, and 2^31 slots (load stays below 0.06) for reference."""
import importlib, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("2048_q-learning_amd")
dev = torch.device("cuda:0")
B = 1 << 20
for cap, launches in ((28, 10), (31, 10)):
    env = pkg.BatchedGame2048Env(B, seed=0, device=dev)
    agent = pkg.BatchedQLearningAgent(1000, learning_rate=0.1, discount_factor=0.99, exploration_rate=0.95,
                                      capacity_log2=cap, seed=0, device=dev)
    keep, agent.epsilon = agent.epsilon, 1.0
    agent.fused_rollout(env, 1024, play_only=True)
    agent.epsilon = keep
    ts, loads = [], []
    for k in range(launches):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); agent.fused_rollout(env, 20); e1.record(); torch.cuda.synchronize()
        ts.append(round(e0.elapsed_time(e1) * 1e3 / 20, 2))
        loads.append(round(agent.stats()["inserts"] / float(1 << cap), 3))
    print(json.dumps({"lib": os.path.basename(pkg._native.LIB_PATH), "cap_log2": cap, "placement": agent.placement["mode"],
                      "us_per_step": ts, "load_after": loads, "drops": agent.stats()["drops"]}), flush=True)
    del agent, env
    torch.cuda.empty_cache()
