#!/usr/bin/env python3
"""Measurement tool: how long does one pass of q2048_table_count / q2048_table_export over the
whole table take (len(q_table), checkpoints)?  One JSON line per capacity."""
import importlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("2048_q-learning_amd")
dev = torch.device("cuda:0")
for cap in (28, 30, 32):
    env = pkg.BatchedGame2048Env(1 << 20, seed=1, device=dev)
    agent = pkg.BatchedQLearningAgent(100, exploration_rate=0.95, capacity_log2=cap, seed=1,
                                      device=dev, placement="plain")
    agent.fused_rollout(env, 64)
    torch.cuda.synchronize()
    rows = agent.table_size()                       # warm
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    rows = agent.table_size()
    e1.record()
    keys, q = agent.export_rows()
    e2.record()
    torch.cuda.synchronize()
    nbytes = 32 << cap
    count_ms = e0.elapsed_time(e1)
    print(json.dumps({"cap_log2": cap, "table_GiB": nbytes / 2 ** 30, "rows": rows,
                      "count_ms": count_ms, "count_GBps": nbytes / count_ms / 1e6,
                      "count_plus_export_rows_ms": e1.elapsed_time(e2), "exported": len(keys)}), flush=True)
    assert len(keys) == rows == agent.stats()["inserts"]
    del agent, env, keys, q
    torch.cuda.empty_cache()
