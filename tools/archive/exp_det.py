#!/usr/bin/env python3
"""Measurement tool: env-steps/s of the deterministic mode (q2048_det_rollout: phase 1, stable
radix partition by (row, action), grouped apply) next to the racing fused kernel on the same job."""
import importlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("2048_q-learning_amd")
if os.environ.get("DET_BITS"):
    pkg._native.use_experiments_build()   # the measurement build: experiment bits 8..23 of `flags`
dev = torch.device("cuda:0")
for B, steps in ((1 << 20, 96), (1 << 16, 256)):
    for mode in ("deterministic", "fused"):
        env = pkg.BatchedGame2048Env(B, seed=0, device=dev)
        agent = pkg.BatchedQLearningAgent(1000, learning_rate=0.1, discount_factor=0.99, exploration_rate=0.95,
                                          capacity_log2=30, seed=0, device=dev)
        agent.fused_rollout(env, 256, play_only=True)      # mid-game boards
        agent.experiment_bits = int(os.environ.get("DET_BITS", "0"), 0) if mode == "deterministic" else 0
        go = agent.deterministic_rollout if mode == "deterministic" else agent.fused_rollout
        go(env, 32)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        go(env, steps)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print(json.dumps({"mode": mode, "B": B, "steps": steps, "us_per_step": round(ms * 1e3 / steps, 2),
                          "env_steps_per_s": B * steps / ms * 1e3, "status": agent.check_status()}), flush=True)
        del agent, env
        torch.cuda.empty_cache()
