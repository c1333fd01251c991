#!/usr/bin/env python3
"""Round 5: the fused rollout's step time against the load of the table it runs on -- ONE 2^CAP-slot table
(default 2^30), filled (untimed) with pseudo-random rows to each load of the list, then 3 x 16 steps of the bench
workload (1 Mi boards, epsilon 0.95) timed with HIP events.  This is the steady state a long run lives in (the
bench's main line runs on a young table); it decides where a growing table should move on.
    python tools/exp_load_curve_prefilled.py [cap_log2=30] [board_size=4] [experiment bits] > profiles/r05_load_curve_prefilled.jsonl
Round 6: every load is also timed with the key set CLOSED (Q2048_FLAG_NO_NEW_ROWS: `frozen_us_per_step`), which is
what a table at its largest capacity runs with once it holds freeze_load of its slots; loads above 0.7 frozen only."""
import importlib
import importlib.util
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
pkg = importlib.import_module("2048_q-learning_amd")
spec = importlib.util.spec_from_file_location("bench", os.path.join(REPO, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

cap = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
bits = [int(x, 0) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]     # experiment bits to alternate between
if any(bits):
    pkg._native.use_experiments_build()
dev, B, S = "cuda:0", 1 << 20, 16
env = pkg.BatchedGame2048Env(B, board_size=n, seed=0, device=dev)
agent = pkg.BatchedQLearningAgent(1000, learning_rate=0.1, discount_factor=0.99, exploration_rate=0.95,
                                  capacity_log2=cap, seed=0, device=dev, board_size=n, placement="chunks",
                                  freeze_load=None)
eps = agent.epsilon
agent.epsilon = 1.0
for _ in range(4):
    agent.fused_rollout(env, 256, play_only=True)           # mid-game boards, as bench.py synthesises them
agent.epsilon = eps
agent.fused_rollout(env, 8)
gen = torch.Generator(device=dev)
gen.manual_seed(1)
words = 1 if n == 4 else 2
chunk = 1 << 25
zeros = torch.zeros((chunk, 4), dtype=torch.float32, device=dev)
for target in (0.0, 0.05, 0.1, 0.15, 0.2, 0.25, 0.3, 0.35, 0.4, 0.45, 0.5, 0.55, 0.6, 0.7, 0.8, 0.9):
    rows = agent.recount_rows()
    want = int(target * (1 << cap)) - rows
    while want > 0:
        k = min(chunk, want)
        keys = torch.randint(-(1 << 62), 1 << 62, (k, words), dtype=torch.int64, device=dev, generator=gen)
        keys |= (-(1 << 63)) if words == 2 else 1
        agent.import_rows_device(keys.view(-1) if words == 1 else keys, zeros[:k])
        want -= k
    before = agent.recount_rows()
    agent.stats(reset=True)
    times, by_bits, frozen_times = [], {}, []
    agent.frozen = True                                   # the key set closed: no row is created at this load
    for r in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        agent.fused_rollout(env, S)
        e1.record()
        e1.synchronize()
        if r:
            frozen_times.append(e0.elapsed_time(e1) * 1e3 / S)
    fst = agent.stats(reset=True)
    assert fst["inserts"] == 0 and (agent.check_status() & ~8) == 0      # (8 = DEEP_ROW: the prefill of loads >= 0.9 places rows beyond the learning probe limit)
    agent.frozen = False
    for r in range(3 * len(bits) if target <= 0.7 else 0):
        agent.experiment_bits = bits[r % len(bits)]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        agent.fused_rollout(env, S)
        e1.record()
        e1.synchronize()
        by_bits.setdefault(hex(agent.experiment_bits), []).append(round(e0.elapsed_time(e1) * 1e3 / S, 2))
        if r % len(bits) == 0:
            times.append(e0.elapsed_time(e1) * 1e3 / S)
    st = agent.stats()
    after = agent.recount_rows()
    print(json.dumps({"cap_log2": cap, "board_size": n, "load_before": before / (1 << cap), "load_after": after / (1 << cap),
                      "us_per_step": [round(t, 2) for t in times], "median_us_per_step": round(sorted(times)[1], 2) if times else None,
                      "frozen_us_per_step": [round(t, 2) for t in frozen_times], "frozen_median_us_per_step": round(sorted(frozen_times)[1], 2),
                      "frozen_drops_per_step": fst["drops"] / max(fst["steps"], 1),
                      "inserts_per_step": st["inserts"] / max(st["steps"], 1), "drops": st["drops"], "by_experiment_bits": by_bits,
                      "algorithmic_frac": (122 if n == 4 else 156) * B / (sorted(times)[1] * 1e-6) / 8e12 if times else None,
                      "frozen_algorithmic_frac": (122 if n == 4 else 156) * B / (sorted(frozen_times)[1] * 1e-6) / 8e12}), flush=True)
