#!/usr/bin/env python3
"""Batch-size and steps-per-launch sweep of the default fused kernel (experiment)."""
import gc, importlib, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("2048_q-learning_amd")
def run(B, S, steps, eps=0.95, cap_log2=29, strict=False):
    env = pkg.BatchedGame2048Env(B, seed=0, device="cuda:0")
    agent = pkg.BatchedQLearningAgent(1000, learning_rate=0.1, discount_factor=0.99, exploration_rate=eps,
                                      capacity_log2=cap_log2, device="cuda:0", strict_td=strict)
    def go(n):
        left = n
        while left > 0:
            k = min(S, left); agent.fused_rollout(env, k); left -= k
    go(64); agent.stats(reset=True); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); go(steps); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1); rate = B * steps / (ms / 1e3)
    print(json.dumps(dict(B=B, S=S, steps=steps, eps=eps, strict=strict, us_per_Mboard_step=round(ms * 1e3 / steps * (1 << 20) / B, 2),
                          steps_per_s=rate, algo_GBs=round(rate * 122 / 1e9, 1))), flush=True)
    del env, agent; gc.collect(); torch.cuda.empty_cache()
for B in (1 << 16, 1 << 18, 1 << 19, 1 << 20, 1 << 21, 1 << 22, 1 << 23):
    run(B, 64, 256 if B <= (1 << 21) else 128, cap_log2=29 if B <= (1 << 21) else 31)
for S in (8, 32, 64, 128, 256):
    run(1 << 20, S, 256)
run(1 << 20, 64, 256, strict=True)
run(1 << 20, 64, 256, eps=0.01)
run(1 << 20, 64, 256, eps=0.01, strict=True)
