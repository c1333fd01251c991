#!/usr/bin/env python3
"""Measurement tool: the unfused single-step kernel (q2048_env_step, 70 B per env-step algorithmic)
alone, for several boards-per-thread settings of the pipelined 4x4 kernel (experiment bits 8..11
of q2048_env_step_ex's flags; 0 = the library's default).  HIP-event time over back-to-back
launches; run it under `rocprofv3 --kernel-trace` for per-dispatch durations (grouped by grid)."""
import importlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("2048_q-learning_amd")
pkg._native.use_experiments_build()   # the measurement build: experiment bits 8..23 of `flags`
N = pkg._native
dev = "cuda:0"
for B, launches in ((1 << 20, 300), (8 << 20, 60)):
    for n in (4, 5):
        for per_thread in ((0, 1, 2, 4, 8) if n == 4 else (0,)):
            env = pkg.BatchedGame2048Env(B, board_size=n, seed=0, device=dev)
            acts = torch.randint(0, 4, (B,), dtype=torch.uint8, device=dev)
            stream = torch.cuda.current_stream().cuda_stream

            def step(t):
                N.check(N.lib().q2048_env_step_ex(
                    env.boards.data_ptr(), env.aux.data_ptr(), acts.data_ptr(), B, n, 0, 0, t,
                    per_thread << 8, None, env._reward.data_ptr(), env._done.data_ptr(),
                    env._max.data_ptr(), env.status.data_ptr(), stream), "env_step_ex")

            for t in range(40):
                step(t)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for t in range(launches):
                step(40 + t)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / launches
            algo = 70 if n == 4 else 25 + 25 + 32 + 6
            print(json.dumps({"kernel": "env_step", "board": n, "B": B, "boards_per_thread": per_thread or "default",
                              "us_per_launch_events": round(us, 2), "algo_GBps_events": round(B * algo / us / 1e3, 1)}), flush=True)
            del env, acts
            torch.cuda.empty_cache()
