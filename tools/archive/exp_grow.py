#!/usr/bin/env python3
"""Measurement tool: what a growth of the Q-table costs, phase by phase (measurement build,
Q2048_DEBUG_GROW=1 prints map / move / count / free to stderr): an empty table grown 2^26 -> 2^32, then a
table driven by 262 144 envs from 2^26 slots on."""
import importlib
import os
import sys
import time

import torch

os.environ["Q2048_DEBUG_GROW"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("2048_q-learning_amd")
pkg._native.use_experiments_build()
dev = torch.device("cuda:0")
agent = pkg.BatchedQLearningAgent(1000, exploration_rate=0.95, capacity_log2="auto", initial_capacity_log2=26,
                                  seed=0, device=dev)
print("empty table, max 2^%d" % agent.max_capacity_log2, flush=True)
while agent.capacity_log2 < min(32, agent.max_capacity_log2):
    t0 = time.perf_counter()
    agent.grow_table()
    print("  host wall %.1f ms" % ((time.perf_counter() - t0) * 1e3), agent.growths[-1], flush=True)
del agent
torch.cuda.empty_cache()
env = pkg.BatchedGame2048Env(1 << 18, seed=0, device=dev)
agent = pkg.BatchedQLearningAgent(1000, exploration_rate=0.95, capacity_log2="auto", initial_capacity_log2=26,
                                  seed=0, device=dev)
print("a table in use", flush=True)
seen = 0
while agent.capacity_log2 < 31:
    agent.fused_rollout(env, 64)
    if len(agent.growths) > seen:
        seen = len(agent.growths)
        print("  ", agent.growths[-1], flush=True)
