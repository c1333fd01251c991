#!/usr/bin/env python3
"""Ablations of the fused kernel's table traffic (experiments)."""
import gc, importlib, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("2048_q-learning_amd")
pkg._native.use_experiments_build()   # the measurement build: experiment bits 8..23 of `flags`
MODES = {"store_plain": 0, "cas": 1, "store_sc1": 2, "add": 6, "none": 4}

def run(name, bits, B=1 << 20, S=16, steps=128, warm=64, eps=0.95, cap_log2=32, warm_bits=None):
    env = pkg.BatchedGame2048Env(B, seed=0, device="cuda:0")
    agent = pkg.BatchedQLearningAgent(1000, learning_rate=0.1, discount_factor=0.99, exploration_rate=eps,
                                      capacity_log2=cap_log2, device="cuda:0")
    def go(n):
        left = n
        while left > 0:
            k = min(S, left); agent.fused_rollout(env, k); left -= k
    keep, agent.epsilon = agent.epsilon, 1.0       # input synthesis as in bench.py: random play, no learner
    for _ in range(4):
        agent.fused_rollout(env, 256, play_only=True)
    agent.epsilon = keep
    agent.experiment_bits = bits if warm_bits is None else warm_bits
    go(warm); agent.stats(reset=True); torch.cuda.synchronize()
    agent.experiment_bits = bits
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); go(steps); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1); st = agent.stats(); rate = B * steps / (ms / 1e3)
    print(json.dumps(dict(name=name, eps=eps, S=S, us_per_step=round(ms * 1e3 / steps, 2), steps_per_s=rate,
                          inserts_per_step=round(st["inserts"] / st["steps"], 3), retries=st["cas_retries"],
                          drops=st["drops"])), flush=True)
    del env, agent; gc.collect(); torch.cuda.empty_cache()

if __name__ == "__main__":
    for eps in (0.95, 0.01):
        for name, m in MODES.items():
            run(name, m << 8, eps=eps)
        run("none+noclaim", (4 << 8) | (1 << 12), eps=eps, warm_bits=0)
        run("none+noclaim+noprobe", (4 << 8) | (1 << 12) | (1 << 13), eps=eps, warm_bits=0)
    for S in (1, 4, 64, 256):
        run(f"store_plain S={S}", 0, S=S, steps=256 if S >= 64 else 128)
    run("cas S=64", 1 << 8, S=64)
    run("store_plain, immediate same-state writes", 1 << 14, S=64, steps=256)
    run("store_plain B=4M", 0, B=4 << 20, steps=64, cap_log2=32)
