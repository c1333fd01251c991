#!/usr/bin/env python3
"""A/B inside one run: 20-step launches of the fused rollout alternating between statistics vectors passed and
NULL on the same env and table (table age drifts equally for both)."""
import importlib, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("2048_q-learning_amd")
dev = torch.device("cuda:0")
B = 1 << 20
env = pkg.BatchedGame2048Env(B, seed=0, device=dev)
agent = pkg.BatchedQLearningAgent(1000, learning_rate=0.1, discount_factor=0.99, exploration_rate=0.95,
                                  capacity_log2=30, seed=0, device=dev)
keep, agent.epsilon = agent.epsilon, 1.0
agent.fused_rollout(env, 1024, play_only=True)
agent.epsilon = keep
agent.fused_rollout(env, 20)
si, sf = agent.stats_i, agent.stats_f
t = {True: [], False: []}
for k in range(24):
    on = k % 2 == 0
    agent.stats_i, agent.stats_f = (si, sf) if on else (None, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); agent.fused_rollout(env, 20); e1.record(); torch.cuda.synchronize()
    t[on].append(e0.elapsed_time(e1) * 1e3 / 20)
agent.stats_i, agent.stats_f = si, sf
for on in (True, False):
    v = t[on]
    print(json.dumps({"statistics": on, "launches": len(v), "us_per_step_mean": round(sum(v) / len(v), 3),
                      "us_per_step": [round(x, 2) for x in v]}))
