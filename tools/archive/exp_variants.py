#!/usr/bin/env python3
"""Throughput sweep of the fused kernel (experiments; prints one line per configuration)."""
import importlib, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("2048_q-learning_amd")

def run(B, S, steps, warm, eps, td_store, cap_log2, independent=False):  # td_store=False -> CAS TD
    env = pkg.BatchedGame2048Env(B, seed=0, device="cuda:0")
    agent = pkg.BatchedQLearningAgent(1000, learning_rate=0.1, discount_factor=0.99, exploration_rate=eps,
                                      capacity_log2=cap_log2, device="cuda:0", strict_td=not td_store,
                                      independent=independent)
    def go(n):
        left = n
        while left > 0:
            k = min(S, left); agent.fused_rollout(env, k); left -= k
    go(warm); agent.stats(reset=True); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); go(steps); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    st = agent.stats()
    rate = B * steps / (ms / 1e3)
    print(json.dumps(dict(B=B, S=S, steps=steps, eps=eps, strict_td=not td_store, cap_log2=cap_log2,
                          us_per_step=ms * 1e3 / steps, steps_per_s=rate, algo_GBs=rate * 122 / 1e9,
                          inserts_per_step=st["inserts"] / st["steps"], cas_retries=st["cas_retries"],
                          drops=st["drops"])), flush=True)
    del env, agent
    torch.cuda.empty_cache()

def env_only(B, steps):
    env = pkg.BatchedGame2048Env(B, seed=0, device="cuda:0")
    acts = torch.randint(0, 4, (B,), dtype=torch.uint8, device="cuda:0")
    for _ in range(8): env.step(acts)
    torch.cuda.synchronize()
    N = pkg._native
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(steps):
        N.lib().q2048_env_step(env.boards.data_ptr(), env.aux.data_ptr(), acts.data_ptr(), B, 4, 0, 0, t,
                               env._reward.data_ptr(), env._done.data_ptr(), env._max.data_ptr(),
                               env.status.data_ptr(), torch.cuda.current_stream().cuda_stream)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    rate = B * steps / (ms / 1e3)
    print(json.dumps(dict(kernel="env_step_only", B=B, us_per_step=ms * 1e3 / steps, steps_per_s=rate,
                          algo_GBs=rate * 70 / 1e9)), flush=True)

if __name__ == "__main__":
    B = 1 << 20
    env_only(B, 200)
    env_only(8 << 20, 50)
    for td_store in (False, True):
        for S in (1, 4, 16, 64):
            run(B, S, 128, 64, 0.95, td_store, 29)
    for td_store in (False, True):
        run(B, 16, 128, 64, 0.01, td_store, 29)
    run(B, 16, 128, 64, 0.95, False, 24)       # table that fits the Infinity Cache... and overflows
    run(B, 16, 128, 64, 0.95, False, 29, independent=True)
    run(4 << 20, 16, 64, 64, 0.95, False, 30)
