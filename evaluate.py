#!/usr/bin/env python3
"""evaluate.py -- play a trained Q-table without learning.

The reference's README lists an `evaluate.py` ("script per la valutazione e il testing",
README.md:52) that its repository does not contain.  This is that script for the MI355X path: it
loads a learner written by `train.py --save`, plays `--episodes` games per env with the stored
values (epsilon-greedy, `--epsilon` 0 by default = the greedy policy of Agent/main.py:38) through
the fused rollout with Q2048_FLAG_NO_LEARN -- rows are read, nothing is created or written -- and
prints one JSON line: games, mean score / return, max-tile histogram.

    python train.py --num-envs 65536 --episodes 40 --save models/q_65536x40.pt
    python evaluate.py --model models/q_65536x40.pt --num-envs 65536 --episodes 2
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def parse_args(argv=None):
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--model", required=True, help="file written by train.py --save")
    p.add_argument("--episodes", type=int, default=1, help="games per env (on average)")
    p.add_argument("--num-envs", type=int, default=65536)
    p.add_argument("--epsilon", type=float, default=0.0, help="exploration while evaluating (0 = greedy)")
    p.add_argument("--seed", type=int, default=12345, help="evaluation draws (spawns); not the training seed")
    p.add_argument("--device", default="cuda")
    p.add_argument("--steps-per-launch", type=int, default=64)
    p.add_argument("--max-steps", type=int, default=100000, help="stop after this many steps per env at the latest")
    p.add_argument("--env-profile", choices=["shaped", "nopenalty"], default="shaped")
    p.add_argument("--reset-shaping-state", action="store_true")
    return p.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    import torch

    pkg = importlib.import_module("2048_q-learning_amd")
    sd = torch.load(args.model, map_location="cpu", weights_only=False)
    n, rows = int(sd["board_size"]), len(sd["q"])
    cap = max(int(sd["capacity_log2"]), 4) if "table" in sd else max(16, (2 * max(rows, 1) - 1).bit_length())
    agent = pkg.BatchedQLearningAgent(1, learning_rate=sd["lr"], discount_factor=sd["gamma"],
                                      exploration_rate=args.epsilon, capacity_log2=cap, seed=args.seed,
                                      device=args.device, board_size=n, placement="plain")
    agent.load_state_dict(sd)
    agent.seed, agent.ctr, agent.epsilon = args.seed, 0, args.epsilon     # evaluation has its own draws
    agent.stats(reset=True)
    env = pkg.BatchedGame2048Env(args.num_envs, n, args.device, args.seed, agent.env_id0,
                                 profile=args.env_profile, reset_shaping_state=args.reset_shaping_state)
    rows_before = agent.table_size()
    target, t0 = args.episodes * args.num_envs, time.time()
    st = agent.stats()
    while st["episodes"] < target and env.ctr < args.max_steps:
        for _ in range(4):
            agent.fused_rollout(env, args.steps_per_launch, learn=False)
        st = agent.stats()
    assert agent.table_size() == rows_before and st["inserts"] == 0      # nothing was learnt
    print(json.dumps({"model": args.model, "rows": rows_before, "board_size": n, "epsilon": args.epsilon,
                      "envs": args.num_envs, "games": st["episodes"], "env_steps": st["steps"],
                      "mean_score": st["mean_score"], "mean_return": st["mean_return"],
                      "valid_move_frac": st["valid_moves"] / max(st["steps"], 1),
                      "max_tile_hist": {str(k): v for k, v in st["max_tile_hist"].items()},
                      "best_tile": max(st["max_tile_hist"], default=0),
                      "seconds": round(time.time() - t0, 3)}))
    return st


if __name__ == "__main__":
    main()
