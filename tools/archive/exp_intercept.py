#!/usr/bin/env python3
"""Measurement tool: what ONE launch of the fused rollout costs on top of its steps, by variant.
launch(S) is timed with HIP events for S = 1..32 on mid-game boards (1 Mi boards, 2^30-slot chunked
table) and fitted as a + b * S; variants: learner-less play (no table traffic at all: a = the launch
itself, boards + aux in and out, ramp and drain), evaluation (probes only), learning with and without
the row cache, and learning without the statistics mirror."""
import importlib
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("2048_q-learning_amd")
dev = torch.device("cuda:0")
B = int(os.environ.get("INTERCEPT_BOARDS", str(1 << 20)))
SS = (1, 2, 4, 8, 16, 32)
def fresh():
    """Mid-game boards and a young table (load 0.02 after the 32 learning steps): every variant starts equal."""
    env = pkg.BatchedGame2048Env(B, seed=0, device=dev)
    agent = pkg.BatchedQLearningAgent(1000, learning_rate=0.1, discount_factor=0.99, exploration_rate=0.95,
                                      capacity_log2=30, seed=0, device=dev)
    agent.fused_rollout(env, 512, play_only=True)
    agent.ctr = env.ctr
    agent.fused_rollout(env, 32)
    torch.cuda.synchronize()
    return env, agent


def timed(env, agent, S, reps, **kw):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in ev:
        e0.record()
        agent.fused_rollout(env, S, **kw)
        e1.record()
    torch.cuda.synchronize()
    return float(np.median([e0.elapsed_time(e1) for e0, e1 in ev])) * 1e3


variants = [("play only (no table)", dict(play_only=True), {}),
            ("evaluation (probe only)", dict(learn=False), {}),
            ("learning, row cache", {}, {}),
            ("learning, no row cache", {}, {"row_cache_enabled": False}),
            ("learning, row cache, NO statistics (stats pointers NULL: no per-block flush, no mirror)", {},
             {"stats_i": None, "stats_f": None}),
            ("learning, row cache (again)", {}, {}),
            ("learning, row cache, NO statistics (again)", {}, {"stats_i": None, "stats_f": None}),
            ("learning, no row cache (again)", {}, {"row_cache_enabled": False})]
ONLY = os.environ.get("INTERCEPT_ONLY")           # run the variants whose name starts with this
for name, kw, attrs in variants:
    if ONLY and not name.startswith(ONLY):
        continue
    env, agent = fresh()
    for k, v in attrs.items():
        setattr(agent, k, v)
    if name.startswith("play"):
        agent.epsilon = 1.0
    us = {S: timed(env, agent, S, 5 if S <= 8 else 3, **kw) for S in SS}     # 183 steps: the table ends at load ~0.14
    b, a = np.polyfit(np.array(SS, dtype=float), np.array([us[S] for S in SS]), 1)
    print(json.dumps({"variant": name, "boards": B, "launch_us": {str(S): round(v, 1) for S, v in us.items()},
                      "fit": {"intercept_us": round(float(a), 1), "per_step_us": round(float(b), 2)},
                      "load": round(agent.table_size() / float(1 << agent.capacity_log2), 4)}), flush=True)
    del env, agent
    torch.cuda.empty_cache()
