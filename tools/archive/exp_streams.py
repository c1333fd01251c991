#!/usr/bin/env python3
"""Measurement tool: does a one-step kernel run faster when its batch is split over several streams, so that
the parts run side by side at different phases (reads of one part against the atomics and streams of another)?
k_q_update (the 4-call API's dominant kernel) on 1 Mi boards, 1 / 2 / 4 / 8 parts."""
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
for parts in (1, 2, 4, 8, 1):
    env = pkg.BatchedGame2048Env(B, seed=0, device=dev)
    agent = pkg.BatchedQLearningAgent(1000, learning_rate=0.1, discount_factor=0.99, exploration_rate=0.95,
                                      capacity_log2=30, seed=0, device=dev)
    agent.fused_rollout(env, 256, play_only=True)
    main = torch.cuda.current_stream()
    streams = [torch.cuda.Stream() for _ in range(parts)]
    cache = agent._cache(B)
    rec = cache.shape[1]
    per = B // parts

    def update(s, a, r, s2, d):
        if parts == 1:
            agent.update_q_value(s, a, r, s2, d)
            return
        fork = torch.cuda.Event(); fork.record(main)
        for k, st in enumerate(streams):
            st.wait_event(fork)
            o = k * per
            N.check(L.q2048_q_update_cached(
                agent.table.data_ptr(), agent.capacity_log2, s.data_ptr() + 16 * o, a.data_ptr() + o,
                r.data_ptr() + 4 * o, s2.data_ptr() + 16 * o, d.data_ptr() + o, per, 4, agent.lr, agent.gamma,
                agent.env_id0 + o, agent.flags, cache.data_ptr() + rec * o, agent.stats_i.data_ptr(),
                agent.status.data_ptr(), st.cuda_stream), "q_update")
            ev = torch.cuda.Event(); ev.record(st); main.wait_event(ev)

    def loop(steps):
        s = env.boards
        for _ in range(steps):
            a = agent.choose_action(s)
            s2, r, d, _ = env.step(a)
            update(s, a, r, s2, d.view(torch.uint8))
            s = env.reset(d)

    loop(8)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); loop(64); e1.record(); torch.cuda.synchronize()
    print(json.dumps({"update_parts": parts, "us_per_step": round(e0.elapsed_time(e1) * 1e3 / 64, 1),
                      "rows": agent.table_size(), "inserts": agent.stats()["inserts"]}), flush=True)
    del agent, env, cache
    torch.cuda.empty_cache()
