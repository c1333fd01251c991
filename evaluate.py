#!/usr/bin/env python3
"""evaluate.py -- play a trained Q-table without learning.

The reference's README lists an `evaluate.py` ("script per la valutazione e il testing",
README.md:52) that its repository does not contain.  This is that script for the MI355X path: it
loads a learner written by `train.py --save`, plays `--episodes` games per env with the stored
values -- rows are read, nothing is created or written -- and prints one JSON line: games, mean
score / return, max-tile histogram.  Two policies:

  --policy legal (default)   argmax of the stored row over the moves that CHANGE the board (the trial-move
        mask of Deep_QLearning/main_dir/mainDQL_CNN_step2.py:168-174, `q2048_legal_moves`; first maximum wins
        as in np.argmax, a state without a row reads as zeros): what a trained table is worth as a player.
        Batched calls: q_values, legal_moves, step, reset(done).
  --policy reference         the agent of Agent/main.py:34-38 with its learning switched off, through the
        fused rollout with Q2048_FLAG_NO_LEARN: argmax over ALL four actions.  The reference's loop relies on
        the update that follows an invalid move (its negative reward sends argmax elsewhere, main.py:43,99);
        with nothing learning, a greedy env repeats an invalid argmax until the >100-repeats rule ends the
        episode (Game2048_env.py:122-127) -- a faithful lr = 0 agent, and a poor measure of the table.

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
    p.add_argument("--policy", choices=["legal", "reference"], default="legal",
                   help="legal: argmax over the moves that change the board; reference: the lr = 0 agent (argmax over all four)")
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
    if args.policy == "legal":
        st = play_legal_moves(torch, agent, env, args, target)
    else:
        st = agent.stats()
        while st["episodes"] < target and env.ctr < args.max_steps:
            for _ in range(4):
                agent.fused_rollout(env, args.steps_per_launch, learn=False)
            st = agent.stats()
    assert agent.table_size() == rows_before and st["inserts"] == 0      # nothing was learnt
    print(json.dumps({"model": args.model, "rows": rows_before, "board_size": n, "epsilon": args.epsilon,
                      "policy": args.policy,
                      "envs": args.num_envs, "games": st["episodes"], "env_steps": st["steps"],
                      "mean_score": st["mean_score"], "mean_return": st["mean_return"],
                      "valid_move_frac": st["valid_moves"] / max(st["steps"], 1),
                      "max_tile_hist": {str(k): v for k, v in st["max_tile_hist"].items()},
                      "best_tile": max(st["max_tile_hist"], default=0),
                      "seconds": round(time.time() - t0, 3)}))
    return st


def play_legal_moves(torch, agent, env, args, target) -> dict:
    """Greedy over the legal moves, batched: per step one row lookup, one legal-move mask, one env step, one masked
    reset; statistics accumulate on the device and are read once per `--steps-per-launch` steps."""
    dev, B = env.device, env.num_envs
    bit = torch.tensor([1, 2, 4, 8], dtype=torch.uint8, device=dev)
    idx = torch.arange(4, device=dev)
    gen = torch.Generator(device=dev).manual_seed(args.seed)
    acc = torch.zeros(5, dtype=torch.float64, device=dev)          # games, score, return, valid moves, steps
    hist = torch.zeros(32, dtype=torch.int64, device=dev)
    aux_f = env.aux.view(torch.float32)
    games = 0
    while games < target and env.ctr < args.max_steps:
        for _ in range(args.steps_per_launch):
            q = agent.q_values(env.boards)
            legal = (env.legal_moves()[:, None] & bit[None, :]) != 0
            best = torch.where(legal, q, torch.full_like(q, float("-inf")))
            top = (best == best.max(dim=1, keepdim=True).values) & legal
            action = ((top.cumsum(1) == 1) & top).to(torch.int64).mul(idx).sum(1)      # the first maximum (np.argmax)
            if args.epsilon > 0.0:
                pick = torch.rand(B, 4, generator=gen, device=dev).masked_fill(~legal, -1.0).argmax(1)
                action = torch.where(torch.rand(B, generator=gen, device=dev) < args.epsilon, pick, action)
            valid = legal.gather(1, action[:, None])[:, 0]            # (no legal move: action 0, the game is over)
            _, _, done, _ = env.step(action.to(torch.uint8))
            d = done.to(torch.float64)
            acc += torch.stack([d.sum(), (env.score.to(torch.float64) * d).sum(), (aux_f[:, 1].to(torch.float64) * d).sum(),
                                valid.sum().to(torch.float64), torch.tensor(float(B), dtype=torch.float64, device=dev)])
            hist += torch.bincount(env.max_log2.to(torch.int64)[done], minlength=32)[:32]
            env.reset(done)
        games = int(acc[0].item())
    a, h = acc.cpu().numpy(), hist.cpu().numpy()
    st = agent.stats()
    st.update({"episodes": int(a[0]), "steps": int(a[4]), "valid_moves": int(a[3]),
               "mean_score": float(a[1] / max(a[0], 1.0)), "mean_return": float(a[2] / max(a[0], 1.0)),
               "max_tile_hist": {1 << k: int(v) for k, v in enumerate(h) if v}})
    return st


if __name__ == "__main__":
    main()
