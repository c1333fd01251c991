#!/usr/bin/env python3
"""bench.py -- env-steps/s of the fused 2048 Q-learning step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (choose -> env.step -> TD update -> reset-on-done,
Agent/main.py:91-101) over the whole batch: 1,048,576 boards per GPU (BASELINE configs[2] at
N = 1; configs[3] = 8 x 1,048,576 at N = 8, weak scaling).  Boards, aux records and the hash
Q-table are resident in HBM before the timed region; the timed region is exactly K steps
(ceil(K / steps_per_launch) launches of the fused kernel) bracketed by barrier + synchronize,
and the reported time is the MAX over ranks.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline      HBM roofline of the dominant kernel (k_fused_rollout): algorithmic bytes per
                launch (122 B per env-step) / average launch duration measured
                with HIP events on the launching stream, against the 8 TB/s HBM3E peak.
  cpu_baseline  the CPU oracle (a C port of the reference loop, oracle/) timed on this host's
                cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

ALGO_BYTES_FUSED_4X4 = 122  # SURVEY.md section 8(d) / BASELINE.md section 4
ALGO_BYTES_FUSED_5X5 = 156
# row-tuple learner: 64 B board+aux stream, 8 gathered 16-B entries (Q(s) and Q(s'); the kernel
# reuses s' as the next s and gathers 4), 4 weight writes, 6 B out.  The 4 MiB weight table is
# cache-resident, so this figure is not HBM traffic.
ALGO_BYTES_ROW_TUPLE = 64 + 8 * 16 + 4 * 4 + 6
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
HBM_COPY_CEILING_GBS = 6290.0


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=256)
    p.add_argument("--warmup", type=int, default=64)
    p.add_argument("--boards-per-gpu", type=int, default=1 << 20)
    p.add_argument("--board-size", type=int, default=4, help="4 (BASELINE configs[2]/[3]) or 5 (configs[4])")
    p.add_argument("--agent", choices=["hash", "row-tuple"], default="hash",
                   help="hash = the reference's whole-board Q-table (headline); row-tuple = "
                        "BASELINE configs[1] flat-array Q (use with --boards-per-gpu 65536)")
    p.add_argument("--steps-per-launch", type=int, default=64,
                   help="env steps per fused launch (boards stay in registers in between)")
    p.add_argument("--eps", type=float, default=0.95)
    p.add_argument("--alpha", type=float, default=0.1)
    p.add_argument("--gamma", type=float, default=0.99)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--cap-log2", type=int, default=0, help="0 = load <= 0.5 and up to half the free device memory")
    p.add_argument("--placement", default="auto",
                   type=lambda v: v if v in ("auto", "plain") else int(v),
                   help="table allocation (agent.place_table): auto | plain (hipMalloc) | "
                        "N (best of N probed candidates)")
    p.add_argument("--strict-td", action="store_true",
                   help="TD write by compare-and-swap loop (Q2048_FLAG_TD_CAS) instead of one store")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget; 0 = skip")
    return p.parse_args()


def pmc_traffic_per_env_step():
    """HBM-side bytes per env-step from the committed rocprofv3 PMC passes (profiles/), or None.
    Collected in separate --pmc passes on the same bench command (tools/pmc_session.sh)."""
    path = os.path.join(REPO, "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return None


def table_capacity_log2(pkg, boards: int, total_steps: int, device) -> int:
    """Every step may create a row: load <= 0.5 at the end of the run, and beyond that half of
    the device's free memory (2^32 slots x 32 B = 128 GiB of the 288 GB): a table that large
    spans the whole memory system, which is where scattered writes run fastest."""
    return pkg.auto_capacity_log2(boards * max(total_steps, 1), device, max_log2=32)


def cpu_baseline(args, seconds: float) -> dict:
    """Times the CPU oracle (kind 'port') on a bounded sample of the same workload."""
    from oracle import oracle as O

    O.lib()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    best = None
    for T in sorted({1, max(1, cores // 4), cores}):
        B, steps = 16384 * T, 24
        envs = O.envs_init(B, 4, args.seed, 0)
        agents = [O.Agent(1000, 4, args.alpha, args.gamma, args.eps) for _ in range(T)]
        for a in agents:
            a.reserve(16384 * 64)      # rows one thread can create in this sample (no rehash)
        O.rollout_mt(envs, agents, 8, args.seed, 0, 0)                 # warm-up
        done, t0 = 0, time.perf_counter()
        budget = seconds / 3
        ctr = 8
        while True:
            O.rollout_mt(envs, agents, steps, args.seed, 0, ctr)
            ctr += steps
            done += B * steps
            dt = time.perf_counter() - t0
            if dt > budget or ctr > 56:
                break
        rate = done / dt
        if best is None or rate > best[0]:
            best = (rate, T, B, ctr - 8, dt)
        del agents, envs
    rate, T, B, steps, dt = best
    return {"value": rate, "unit": "env-steps/s", "cores": T, "kind": "port",
            "sample": f"oracle/q2048_oracle.c orc_rollout_mt: {B} boards x {steps} steps, "
                      f"{T} thread(s) with one private Q-table each, {dt:.1f} s "
                      f"(host has {cores} usable cores)",
            "reference_python_1core_survey_container": 11144.0}


def main():
    args = parse_args()
    if args.steps < 1 or args.warmup < 0:
        raise SystemExit("--steps must be >= 1 and --warmup >= 0")
    pkg = importlib.import_module("2048_q-learning_amd")
    rank, local_rank, world = pkg.dist.init_process_group()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torchrun "
                         f"--nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible")
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())  # 1 rank = 1 GPU on a node
    torch.cuda.set_device(dev)

    B = args.boards_per_gpu
    shard = pkg.weak_shard(B, world, rank)
    S = max(1, min(args.steps_per_launch, args.steps))
    cap_log2 = args.cap_log2 or table_capacity_log2(pkg, B, args.steps + args.warmup, dev)

    algo_bytes = ALGO_BYTES_FUSED_4X4 if args.board_size == 4 else ALGO_BYTES_FUSED_5X5
    if args.agent == "row-tuple":
        algo_bytes = ALGO_BYTES_ROW_TUPLE
    env = pkg.BatchedGame2048Env(shard.num_envs, board_size=args.board_size, seed=args.seed,
                                 env_id0=shard.env_id0, device=dev)
    if args.agent == "row-tuple":
        agent = pkg.BatchedRowTupleAgent(1000, learning_rate=args.alpha, discount_factor=args.gamma,
                                         exploration_rate=args.eps, seed=args.seed,
                                         env_id0=shard.env_id0, device=dev)
    else:
        agent = pkg.BatchedQLearningAgent(1000, learning_rate=args.alpha, discount_factor=args.gamma,
                                          exploration_rate=args.eps, capacity_log2=cap_log2,
                                          seed=args.seed, env_id0=shard.env_id0, device=dev,
                                          strict_td=args.strict_td, board_size=args.board_size,
                                          placement=args.placement)

    def run(steps):
        launches = 0
        left = steps
        while left > 0:
            k = min(S, left)
            agent.fused_rollout(env, k)
            left -= k
            launches += 1
        return launches

    run(args.warmup)                       # mid-game boards, warm table (untimed)
    pkg.dist.allreduce_stats(agent.stats_i, agent.stats_f)   # untimed: RCCL builds its rings lazily
    agent.stats(reset=True)
    torch.cuda.synchronize(dev)
    pkg.dist.barrier()
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    launches = run(args.steps)             # exactly K steps
    ev1.record()
    pkg.dist.allreduce_stats(agent.stats_i, agent.stats_f)   # the path's only collective
    torch.cuda.synchronize(dev)
    pkg.dist.barrier()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1)      # HIP events on the launching stream
    wall_max = pkg.dist.max_over_ranks(wall, device=dev)
    kernel_ms_max = pkg.dist.max_over_ranks(kernel_ms, device=dev)

    st = agent.stats()                     # all-reduced: whole-job numbers
    table_rows = agent.table_size() if args.agent == "hash" else None   # this rank's replica, after W + K steps
    status = agent.check_status()
    total_env_steps = shard.total_envs * args.steps
    assert st["steps"] == total_env_steps, (st["steps"], total_env_steps)
    value = total_env_steps / wall_max

    # roofline of the dominant kernel, per launch, on this rank
    avg_launch_s = (kernel_ms_max / 1e3) / launches
    algo_bytes_per_launch = algo_bytes * shard.num_envs * (args.steps / launches)
    achieved = algo_bytes_per_launch / avg_launch_s / 1e9
    roofline = {"bound": "hbm", "kernel": "k_rt_fused_rollout" if args.agent == "row-tuple" else "k_fused_rollout", "achieved": achieved,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "frac_of_measured_copy_ceiling": achieved / HBM_COPY_CEILING_GBS,
                "traffic": None,  # filled below from the committed PMC passes
                "algorithmic_bytes_per_env_step": algo_bytes,
                "avg_launch_ms": avg_launch_s * 1e3, "launches": launches,
                "note": f"register-resident, K={S} env steps per launch: boards/aux cross HBM once "
                        f"per launch, the figure counts them once per step (SURVEY 8(d))"}

    pmc = pmc_traffic_per_env_step() if (args.board_size == 4 and args.agent == "hash") else None
    if pmc is not None:
        roofline["traffic"] = pmc["bytes_per_env_step"] * shard.num_envs * (args.steps / launches)
        roofline["traffic_source"] = pmc["source"]
        roofline["traffic_bytes_per_env_step"] = pmc["bytes_per_env_step"]
    out = {
        "metric": "env_steps_per_sec", "value": value, "unit": "env-steps/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall_max * 1e3 / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": f"{B} parallel {args.board_size}x{args.board_size} boards per GPU (uint8 "
                               f"log2), {'device open-addressed hash Q-table' if args.agent == 'hash' else 'flat-array row-tuple Q (4 MiB)'}, fused step+select+TD kernel "
                               f"(BASELINE configs[{(2 if args.board_size == 4 else 4) if args.agent == 'hash' else 1}]"
                               f"{'' if world == 1 else '/[3] sharded, one Q replica per GPU'})",
                   "boards_per_gpu": B, "total_boards": shard.total_envs,
                   "steps_per_launch": S, "table_capacity_log2": cap_log2,
                   "table_bytes_per_gpu": (1 << cap_log2) * 32, "epsilon": args.eps,
                   "alpha": args.alpha, "gamma": args.gamma, "seed": args.seed,
                   "td_write": "compare-and-swap" if args.strict_td else "store (last writer wins)",
                   "table_placement": getattr(agent, "placement", None),
                   "parallelism": f"env-batch x{world}, RCCL all-reduce of statistics only"},
        "roofline": roofline,
        "stats": {"episodes": st["episodes"], "mean_return": st["mean_return"],
                  "mean_score": st["mean_score"], "valid_move_frac": st["valid_moves"] / max(st["steps"], 1),
                  "table_inserts": st["inserts"], "drops": st["drops"],
                  "table_rows_per_gpu": table_rows, "claim_timeouts": pkg._native.claim_timeouts(),
                  "table_load_factor": None if table_rows is None else table_rows / float(1 << cap_log2),
                  "cas_retries": st["cas_retries"], "status": status,
                  "max_tile_hist": {str(k): v for k, v in st["max_tile_hist"].items()}},
        "kernel_ms_total": kernel_ms_max,
    }
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
