#!/usr/bin/env python3
"""Measurement tool: the reference's loop body through the batched 4-call API (choose_action ->
step -> update_q_value -> reset(done): four launches per env-step, driven from Python) next to the
fused rollout, same job."""
import importlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("2048_q-learning_amd")
TD_BITS = int(os.environ.get("UNFUSED_TD_BITS", "0"), 0)      # measurement build: write mode of k_q_update (2 << 8 = sc1)
if TD_BITS:
    pkg._native.use_experiments_build()
N_BOARD = int(os.environ.get("UNFUSED_BOARD_SIZE", "4"))     # 5: the 5x5 geometry
dev = torch.device("cuda:0")
for B in (1 << 20, 1 << 16):
    env = pkg.BatchedGame2048Env(B, board_size=N_BOARD, seed=0, device=dev)
    agent = pkg.BatchedQLearningAgent(1000, learning_rate=0.1, discount_factor=0.99, exploration_rate=0.95,
                                      capacity_log2=32 if B == 1 << 20 else 28, seed=0, device=dev, placement="plain",
                                      board_size=N_BOARD)
    agent.fused_rollout(env, 256, play_only=True)
    agent.flags |= TD_BITS

    def loop(steps):                      # Agent/main.py:92-100, :81 in batched form
        s = env.boards
        for _ in range(steps):
            a = agent.choose_action(s)
            s2, r, d, _ = env.step(a)     # writes the other board buffer: `s` stays intact, no copy
            agent.update_q_value(s, a, r, s2, d)
            s = env.reset(d)

    keep_stats = agent.stats_i
    if os.environ.get("UNFUSED_NO_STATS"):          # how much of k_q_update is its per-block statistics atomics?
        agent.stats_i = None
    loop(8)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    steps = 64
    e0.record(); loop(steps); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(json.dumps({"api": "4-call (choose, step, update, reset)", "B": B, "board": N_BOARD, "steps": steps,
                      "us_per_step": round(ms * 1e3 / steps, 1), "env_steps_per_s": B * steps / ms * 1e3}), flush=True)
    agent.ctr = env.ctr
    agent.stats_i = keep_stats
    agent.flags &= 0xff
    e0.record(); agent.fused_rollout(env, steps); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(json.dumps({"api": "fused_rollout", "B": B, "steps": steps, "us_per_step": round(ms * 1e3 / steps, 1),
                      "env_steps_per_s": B * steps / ms * 1e3}), flush=True)
    del agent, env
    torch.cuda.empty_cache()
