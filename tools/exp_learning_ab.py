#!/usr/bin/env python3
"""Measurement tool: does the way concurrent TD updates reach the shared table change what is
learnt?  The reference's experiment size (200 000 games, plots/summary_statistics_cleaned.csv:2-5)
with 4096 envs on one table, epsilon schedule of Agent/main.py:45-57 applied once per epoch
(one episode per env on average), for every write mode:
  store/64   the default: one 4-byte store, 64 env steps per launch (a store may sit in the
             writing XCD's L2 until the launch ends)
  store/1    the same with one step per launch (every update is visible before the next step)
  sc1        write-through store (agent scope)
  cas        compare-and-swap loop: concurrent updates of one entry serialise (Q2048_FLAG_TD_CAS)
  det        deterministic mode: updates of a step grouped by (state, action), applied in env order
  frozen     the default store mode on a table too small for the run (2^--frozen-capacity-log2 slots): it closes its
             key set at freeze_load (Q2048_FLAG_NO_NEW_ROWS, round 6) and the rest of the run learns on the rows it has
  det-frozen the same table in deterministic mode: after the freeze the envs' visit rows live in the row cache
             (q2048_det_rollout_cached)
One JSON line per (mode, seed): mean return / score and max-tile histogram of the last 10 epochs."""
import argparse
import importlib
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("2048_q-learning_amd")

p = argparse.ArgumentParser()
p.add_argument("--num-envs", type=int, default=4096)
p.add_argument("--episodes", type=int, default=50)
p.add_argument("--seeds", type=int, default=3)
p.add_argument("--modes", default="store/64,store/1,sc1,cas,det")
p.add_argument("--capacity-log2", type=int, default=27)
p.add_argument("--frozen-capacity-log2", type=int, default=25)
p.add_argument("--last", type=int, default=10, help="epochs at the end of the run that are summarised")
args = p.parse_args()
if any(m in ("sc1",) for m in args.modes.split(",")):
    pkg._native.use_experiments_build()   # the measurement build: experiment bits 8..23 of `flags`
dev = torch.device("cuda:0")
B, E = args.num_envs, args.episodes
MODES = {"store/64": dict(S=64), "store/1": dict(S=1), "sc1": dict(S=64, bits=0x200),
         "cas": dict(S=64, strict=True), "det": dict(S=64, det=True), "frozen": dict(S=64, frozen=True),
         "det-frozen": dict(S=64, det=True, frozen=True)}
for mode in args.modes.split(","):
    cfg = MODES[mode]
    for seed in range(args.seeds):
        env = pkg.BatchedGame2048Env(B, seed=seed, device=dev)
        agent = pkg.BatchedQLearningAgent(E, learning_rate=0.1, discount_factor=0.99, exploration_rate=0.95,
                                          capacity_log2=args.frozen_capacity_log2 if cfg.get("frozen") else args.capacity_log2,
                                          seed=seed, device=dev, strict_td=cfg.get("strict", False), placement="plain",
                                          freeze_load=0.5 if cfg.get("frozen") else None)
        agent.experiment_bits = cfg.get("bits", 0)
        S, t0 = cfg["S"], time.time()
        epoch, total_eps, per_epoch, drops = 0, 0, [], 0
        chunk = 64 // S
        while epoch < E:
            for _ in range(chunk):
                if cfg.get("det"):
                    agent.deterministic_rollout(env, S)
                else:
                    agent.fused_rollout(env, S)
            st = agent.stats(reset=True)
            total_eps += st["episodes"]
            drops += st["drops"]
            cur = {"episodes": st["episodes"], "return_sum": st["return_sum"], "score_sum": st["score_sum"],
                   "hist": st["max_tile_hist"]}
            if per_epoch and len(per_epoch) > epoch:
                e = per_epoch[epoch]
                e["episodes"] += cur["episodes"]; e["return_sum"] += cur["return_sum"]; e["score_sum"] += cur["score_sum"]
                for k, v in cur["hist"].items():
                    e["hist"][k] = e["hist"].get(k, 0) + v
            else:
                per_epoch.append(cur)
            while epoch < total_eps // B and epoch < E:
                agent.decay_exploration(epoch)
                epoch += 1
        torch.cuda.synchronize()
        last = per_epoch[-args.last:]
        n = sum(e["episodes"] for e in last)
        hist = {}
        for e in last:
            for k, v in e["hist"].items():
                hist[k] = hist.get(k, 0) + v
        print(json.dumps({"mode": mode, "seed": seed, "envs": B, "epochs": E, "games": total_eps,
                          "env_steps": env.ctr * B, "seconds": round(time.time() - t0, 2),
                          "last10_games": n, "last10_mean_return": sum(e["return_sum"] for e in last) / n,
                          "last10_mean_score": sum(e["score_sum"] for e in last) / n,
                          "last10_max_tile_hist": {str(k): v for k, v in sorted(hist.items())},
                          "first5_mean_return": sum(e["return_sum"] for e in per_epoch[:5]) / max(1, sum(e["episodes"] for e in per_epoch[:5])),
                          "table_rows": agent.table_size(), "epsilon_end": agent.epsilon, "drops": drops,
                          "frozen_at": agent.frozen_at,
                          "status": agent.check_status()}), flush=True)
        del agent, env
        torch.cuda.empty_cache()
