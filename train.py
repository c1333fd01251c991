#!/usr/bin/env python3
"""train.py -- the episode loop of the reference (QLearningBase/Agent/main.py:65-115) on MI355X.

The reference's README advertises `python train.py --episodes N --alpha A --gamma G --epsilon E`
(README.md:62-75) but ships no such file: its training loop is the `__main__` block of
Agent/main.py with every hyper-parameter a literal (:67-68).  This script is that loop with the
README flags, in two modes:

  --num-envs 1   the reference loop body, line for line, on the drop-in adapters
                 (Game2048_env / QLearningAgent with the reference's Python types), one CSV row
                 per finished episode with the reference's header (Agent/main.py:71-76).
  --num-envs B   B boards per GPU through the fused rollout kernel (`--steps-per-launch` env
                 steps per launch); epsilon decays once per `B` finished episodes (one "epoch" =
                 one episode per env on average); one CSV row per report interval with the
                 aggregated statistics.  Under torchrun every rank trains its own shard of the
                 env batch and its own Q-table replica; only the statistics are all-reduced.
"""
from __future__ import annotations

import argparse
import csv
import importlib
import importlib.util
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def parse_args(argv=None):
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--episodes", type=int, default=1000, help="total episodes (per env on average)")
    p.add_argument("--alpha", type=float, default=0.1, help="learning rate (README.md:66)")
    p.add_argument("--gamma", type=float, default=0.99, help="discount factor (README.md:67)")
    p.add_argument("--epsilon", type=float, default=0.95, help="initial exploration rate (README.md:68)")
    p.add_argument("--epsilon-min", type=float, default=0.01)
    p.add_argument("--num-envs", type=int, default=1, help="boards per GPU")
    p.add_argument("--gpus", type=int, default=1,
                   help="GPUs of this node: with N > 1 and no process group in the environment the script "
                        "starts its own N ranks (one per GPU); under torchrun it joins the given group")
    p.add_argument("--board-size", type=int, default=4)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--device", default="cuda",
                   help='"cuda[:i]" (MI355X, the HIP library) or "cpu": the CPU twin (libq2048_host.so: the same C ABI '
                        "compiled for the host from the kernels' own per-lane arithmetic) -- an explicit device, never "
                        "a fallback")
    p.add_argument("--steps-per-launch", type=int, default=64)
    p.add_argument("--capacity-log2", type=int, default=0,
                   help="Q-table slots = 2^n, fixed; 0 (default) = a table that grows like the reference's "
                        "defaultdict, without stopping the loop: it starts at --initial-capacity-log2 slots and "
                        "grows fourfold whenever 0.35 of it is in use (the next table is mapped by a host thread "
                        "while the rollouts go on; the rows move between two launches)")
    p.add_argument("--initial-capacity-log2", type=int, default=0,
                   help="first capacity of the growing table; 0 (default) = 2^30 slots (32 GiB) when that is at most "
                        "an eighth of the free device memory, else the largest power of two that is")
    p.add_argument("--freeze-load", type=float, default=0.5,
                   help="what a table does when it cannot grow any more (--capacity-log2, or the growing table at the "
                        "largest capacity the device holds): once this share of its slots holds rows it takes no new "
                        "rows -- rows that exist keep learning, a state without a row reads as zeros (the defaultdict's "
                        "fresh row, Agent/main.py:16) and its updates are dropped and counted in the Drops column "
                        "(SURVEY 7.3; Q2048_FLAG_NO_NEW_ROWS).  0 = never: the table fills up and slows down by orders "
                        "of magnitude")
    p.add_argument("--growth", choices=["async", "sync"], default="async",
                   help="async (default): growth off the critical path (q2048_table_grow_begin / _commit / _finish); "
                        "sync: the host-synchronous q2048_table_grow (same table, bit for bit)")
    p.add_argument("--verify-every", type=int, default=0,
                   help="batched mode: every N reports also count the table's occupied slots against the rows the "
                        "kernels created (one streaming pass, synchronising).  0 (default): the check runs where it "
                        "is free or needed -- at every growth (on the table left behind), before --save and at the "
                        "end of the run")
    p.add_argument("--strict-td", action="store_true", help="compare-and-swap TD writes (bounded: after 16 lost "
                   "races an update is stored plainly and counted in the statistics)")
    p.add_argument("--deterministic", action="store_true",
                   help="batched mode: reproducible two-phase steps (slower; see deterministic_rollout)")
    p.add_argument("--agent", choices=["hash", "row-tuple"], default="hash",
                   help="hash = the reference's whole-board Q-table; row-tuple = BASELINE configs[1]")
    p.add_argument("--log", default="debug_log.csv", help="CSV log path (Agent/main.py:71)")
    p.add_argument("--report-every", type=int, default=10, help="launches between CSV rows (batched)")
    p.add_argument("--max-steps", type=int, default=0, help="stop after this many env steps per env (0 = off)")
    p.add_argument("--episode-log", default="", help="batched mode: also write one row per finished "
                   "episode with the reference's columns (Agent/main.py:71-76) + Env to this CSV")
    p.add_argument("--reset-shaping-state", action="store_true",
                   help="opt-in fix of a reference bug: Game2048_env.reset (Game2048_env.py:187-191) keeps "
                        "previous_max and the consecutive-action streak, so an episode ended by the "
                        ">100-repeats rule ends again on its first repeated action; with this flag a reset "
                        "restores them.  Default: the reference's behaviour, bit for bit")
    p.add_argument("--env-profile", choices=["shaped", "nopenalty"], default="shaped",
                   help="shaped = QLearningBase's Game2048_env (the reference's tabular path); nopenalty = "
                        "the DQN path's env (Deep_QLearning/environment/Game2048_nopenalty_env.py: reward = "
                        "merge score or -10, done = game over)")
    p.add_argument("--save", default="", help="write the trained learner (Q-table rows, epsilon schedule, counters) "
                   "to this file when training ends -- the models/ directory of the reference's README; "
                   "evaluate.py and --resume read it")
    p.add_argument("--resume", default="", help="continue from a file written by --save (same agent arguments): "
                   "the table, the statistics, the epoch counter, epsilon where the saved run left it (the "
                   "schedule's phase limits are rebuilt from this run's --episodes) and, when the batch has "
                   "the saved shape, the boards -- save at epoch k + resume to N trains what N epochs train")
    p.add_argument("--launch-timeout", type=float, default=86400.0,
                   help="self-launched ranks (--gpus N > 1 outside torchrun) are stopped after this many seconds")
    p.add_argument("--stop-epoch", type=int, default=0, help="stop (and --save) once this many epochs of the "
                   "--episodes schedule are done; 0 = run to the end")
    p.add_argument("--summary", default="", help="after the run, write the per-episode log's summary row "
                   "(layout of the reference's plots/summary_statistics_cleaned.csv) to this CSV")
    return p.parse_args(argv)


def log_debug_info(file_path, episode, action, q_values, reward, total_reward, max_value):
    """Agent/main.py:59-62."""
    with open(file_path, mode="a", newline="") as file:
        csv.writer(file).writerow([episode, action, q_values, reward, total_reward, max_value])


def train_single(args, pkg):
    """Agent/main.py:65-115 with the adapters: the loop body below is the reference's."""
    # (play_first_board: the reset before the first episode keeps the constructor's game, so that this loop plays
    # what lane 0 of the batched rollout plays and what the golden transcript G6 recorded -- draw for draw)
    env = pkg.Game2048_env(device=args.device, seed=args.seed, profile=args.env_profile,  # :66
                           reset_shaping_state=args.reset_shaping_state, play_first_board=not args.resume)
    num_episodes = args.episodes                                                       # :67
    agent = pkg.QLearningAgent(num_episodes, action_space=env.action_space.n,          # :68
                               learning_rate=args.alpha, discount_factor=args.gamma,
                               exploration_rate=args.epsilon, exploration_min=args.epsilon_min,
                               capacity_log2=args.capacity_log2 or 22, device=args.device,
                               seed=args.seed)
    first_episode = 0
    if args.resume:
        sd = _load(args.resume, 0)
        agent._b.load_state_dict(sd)
        first_episode = _resume_schedule(agent._b, sd, args, 1)
    log_file = args.log                                                                # :71
    with open(log_file, mode="w", newline="") as file:                                 # :74-76
        csv.writer(file).writerow(["Episode", "Action", "Q-Values", "Reward", "Total-Reward", "Max Value"])
    max_number_in_train = 0
    t0, steps = time.time(), 0
    stop = min(num_episodes, args.stop_epoch) if args.stop_epoch else num_episodes
    agent._b.train_progress = {"epoch": stop}
    for episode in range(first_episode, stop):                                         # :80
        state = env.reset()                                                            # :81
        state = tuple(map(tuple, state))                                               # :82
        done = False
        total_reward = 0
        max_number_in_train = max(max_number_in_train, int(np.max(env.game.board)))    # :85-86
        while not done:                                                                # :91
            action = agent.choose_action(state)                                        # :92
            next_state, reward, done, info = env.step(action)                          # :93
            next_state = tuple(map(tuple, next_state))                                 # :94
            agent.update_q_value(state, action, reward, next_state, done)              # :99
            q_values = agent.q_table[state]                                            # :96 (post-update alias)
            max_value = np.max(next_state)                                             # :97
            state = next_state                                                         # :100
            total_reward += reward                                                     # :101
            steps += 1
            if done:                                                                   # :103-105
                log_debug_info(log_file, episode, action, q_values, reward, total_reward, max_value)
        agent.decay_exploration(episode)                                               # :109
        if episode % 100 == 0:
            print(f"Episode {episode}: Total Reward: {total_reward} epsilon {agent.epsilon:.4f} "
                  f"rows {len(agent.q_table)} steps/s {steps / (time.time() - t0):.0f}")
    return agent


def train_batched(args, pkg):
    import torch

    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and args.device != "cpu":
        # a rank of a multi-GPU job sits on the CPUs next to ITS GPU -- the device it will really use: the index named
        # by --device cuda:N, else one GPU per local rank (sysfs + sched_setaffinity, no GPU call yet;
        # torch.cuda.device_count() does not initialise the GPU)
        idx = int(args.device.split(":", 1)[1]) if ":" in args.device else (
            int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
        pkg.launch.pin_to_gpu_numa_node(idx)
    rank, local_rank, world = pkg.dist.init_process_group()
    if args.device == "cpu":
        dev = torch.device("cpu")
    elif ":" in args.device:
        dev = torch.device(args.device)
    else:       # one rank = one GPU of the node (ranks share GPUs only in rehearsals on a smaller box)
        dev = torch.device("cuda", local_rank % max(torch.cuda.device_count(), 1))
    if dev.type == "cuda":
        torch.cuda.set_device(dev)
    shard = pkg.weak_shard(args.num_envs, world, rank)
    B = shard.num_envs
    # the reference's q_table is a defaultdict (Agent/main.py:16): no capacity to choose.  Without
    # --capacity-log2 the device table grows as the run goes (BatchedQLearningAgent(capacity_log2="auto"))
    cap = args.capacity_log2 or "auto"
    env = pkg.BatchedGame2048Env(B, args.board_size, dev, args.seed, shard.env_id0,
                                 profile=args.env_profile, reset_shaping_state=args.reset_shaping_state)
    reducer = pkg.StatsAllReduce(dev)                 # the only collective, on a stream of its own
    if args.agent == "row-tuple":
        agent = pkg.BatchedRowTupleAgent(args.episodes, 4, args.alpha, args.gamma, args.epsilon,
                                         args.epsilon_min, dev, args.seed, shard.env_id0)
    else:
        agent = pkg.BatchedQLearningAgent(args.episodes, 4, args.alpha, args.gamma, args.epsilon,
                                          args.epsilon_min, cap, dev, args.seed, shard.env_id0,
                                          strict_td=args.strict_td, board_size=args.board_size,
                                          initial_capacity_log2=args.initial_capacity_log2 or "auto",
                                          async_growth=args.growth == "async", freeze_load=args.freeze_load or None)
    if args.deterministic and args.agent != "hash":
        raise SystemExit("--deterministic applies to the hash-table agent")
    if (args.save or args.resume) and args.agent != "hash":
        raise SystemExit("--save / --resume apply to the hash-table agent")
    epoch0 = 0
    if args.resume:
        sd = _load(args.resume, rank)
        agent.load_state_dict(sd)
        epoch0 = _resume_schedule(agent, sd, args, shard.total_envs)
        env_sd = sd.get("env")
        if env_sd is not None and tuple(env_sd["boards"].shape) == tuple(env.boards.shape) and \
                env_sd["env_id0"] == shard.env_id0 and env_sd["board_size"] == args.board_size:
            env.load_state_dict(env_sd)           # the saved games go on where they stopped
        else:
            env.ctr = agent.ctr                   # fresh boards, the learner's draw counter
    if rank == 0:
        with open(args.log, mode="w", newline="") as fh:
            csv.writer(fh).writerow(["Epoch", "Episodes", "Env-Steps", "Epsilon", "Mean-Return",
                                     "Mean-Score", "Max Value", "Table-Rows", "Drops", "Steps/s"])
    ep_log = None
    if args.episode_log and (args.agent != "hash" or args.deterministic):
        raise SystemExit("--episode-log needs the fused hash-table path")
    if args.episode_log:
        ep_log = pkg.EpisodeLog(max(4 * B, 1 << 16), device=dev)
        ep_path = args.episode_log if world == 1 else f"{args.episode_log}.rank{rank}"
        with open(ep_path, mode="w", newline="") as fh:
            csv.writer(fh).writerow(["Episode", "Action", "Q-Values", "Reward", "Total-Reward",
                                     "Max Value", "Env"])
    # a resumed run goes on from the saved statistics: `epoch` epochs are done, epsilon has been
    # decayed that many times (Agent/main.py:109), the episode count is the restored one
    if args.agent == "hash" and getattr(agent, "_growth", None) is not None:
        # set-up, before the clock starts: the table the first growth moves into is mapped (a host thread of the
        # library has been at it since the agent was built); every later one is mapped while the run goes on
        ms = agent.wait_for_prefetch()
        print(f"[rank {rank}] next table (2^{agent._growth.new_capacity_log2} slots) "
              + (f"mapped in {ms:.0f} ms before the run" if agent._growth.prepared_ok else
                 f"could NOT be mapped ({ms:.0f} ms): the run stays on 2^{agent.capacity_log2} slots"), flush=True)
    total_eps, epoch, launches, t0 = 0, epoch0, 0, time.time()
    if args.resume:
        reducer.start(agent.stats_i, agent.stats_f)
        total_eps = pkg.stats_dict(*reducer.wait())["episodes"]
    stop_epoch = min(args.episodes, args.stop_epoch) if args.stop_epoch else args.episodes
    target = stop_epoch * shard.total_envs
    best_tile, grown, reports, said_frozen = 0, 0, 0, False
    agent.train_progress = {"epoch": epoch}
    agent.train_env = env
    while total_eps < target:
        if args.deterministic:
            agent.deterministic_rollout(env, args.steps_per_launch)
        elif ep_log is not None:
            agent.fused_rollout(env, args.steps_per_launch, episode_log=ep_log)
        else:
            agent.fused_rollout(env, args.steps_per_launch)
        launches += 1
        if ep_log is not None:
            with open(ep_path, mode="a", newline="") as fh:      # log_debug_info, Agent/main.py:59-62
                wr = csv.writer(fh)
                for rec in ep_log.drain():
                    wr.writerow(pkg.EpisodeLog.csv_row(rec) + [int(rec["env_id"])])
        if launches % args.report_every and not args.max_steps:
            continue
        reducer.start(agent.stats_i, agent.stats_f)       # a few hundred bytes, summed over the ranks
        st = pkg.stats_dict(*reducer.wait())
        total_eps = st["episodes"]
        while epoch < total_eps // shard.total_envs and epoch < args.episodes:
            agent.decay_exploration(epoch)                # Agent/main.py:109, once per epoch
            epoch += 1
        agent.train_progress = {"epoch": epoch}
        best_tile = max(st["max_tile_hist"], default=0)
        reports += 1
        if args.agent == "hash":
            # occupied slots == rows the kernels created: every growth checks the table it leaves behind (no extra
            # pass); a full count only when asked for (--verify-every), before --save and at the end of the run
            if args.verify_every and reports % args.verify_every == 0:
                agent.verify_table()
            grown = _report_growths(agent, grown, rank)
            if agent.frozen and not said_frozen:
                said_frozen, f = True, agent.frozen_at
                print(f"[rank {rank}] table frozen at step {f['at_step']}: 2^{f['capacity_log2']} slots hold {f['rows']} "
                      f"rows (load {f['load']:.3f} >= --freeze-load {agent.freeze_load}); no new rows from here on, "
                      "updates of states without a row are dropped and counted", flush=True)
        if rank == 0:
            rate = st["steps"] / (time.time() - t0)
            with open(args.log, mode="a", newline="") as fh:
                csv.writer(fh).writerow([epoch, total_eps, st["steps"], f"{agent.epsilon:.6f}",
                                         f"{st['mean_return']:.4f}", f"{st['mean_score']:.1f}", best_tile,
                                         st["inserts"], st["drops"], f"{rate:.4e}"])
            print(f"epoch {epoch}/{args.episodes} episodes {total_eps} steps {st['steps']} eps "
                  f"{agent.epsilon:.4f} mean return {st['mean_return']:.3f} best tile {best_tile} "
                  f"rows {st['inserts']} drops {st['drops']} {rate:.3e} env-steps/s", flush=True)
        if args.max_steps and env.ctr >= args.max_steps:
            break
    agent.check_status()
    if args.agent == "hash":
        check = agent.verify_table()                      # (waits for a growth still in flight and checks it too)
        grown = _report_growths(agent, grown, rank)
        print(f"[rank {rank}] table check passed: {check['rows']} rows == rows created, 2^{check['capacity_log2']} "
              f"slots, load {check['load']:.3f}" + (" (frozen)" if agent.frozen else ""), flush=True)
    # the collectives are over: leave the group in order (a rank that simply exits lets the backend's threads be torn
    # down under it -- an 8-rank gloo job once ended a rank with "terminate called without an active exception")
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        pkg.dist.barrier()
        torch.distributed.destroy_process_group()
    return agent


def _report_growths(agent, grown, rank) -> int:
    while len(agent.growths) > grown:
        g = agent.growths[grown]
        grown += 1
        print(f"[rank {rank}] table grew 2^{g['from_log2']} -> 2^{g['to_log2']} slots at step {g['at_step']}: "
              f"{g['rows']} rows moved in {g['ms']:.1f} ms on the stream"
              + (f", host blocked {g['host_ms']:.1f} ms (the table had been mapped in {g.get('prepare_ms', 0.0):.0f} ms "
                 "by the library's host thread)" if "host_ms" in g else " (host-synchronous)"), flush=True)
    return grown


def _resume_schedule(agent, sd, args, total_envs) -> int:
    """After load_state_dict: how many epochs the saved run had finished, and the epsilon schedule
    of THIS run (phase limits and decay rates from --episodes / --epsilon / --epsilon-min, as the
    constructor computes them, Agent/main.py:25-32) carrying on from the saved epsilon -- the saved
    limits belong to the saved run's --episodes and would pin a longer run to epsilon_min."""
    eps_now = agent.schedule.epsilon
    agent.schedule.__init__(args.episodes, args.epsilon, args.epsilon_min)
    agent.schedule.epsilon = eps_now
    agent.total_epochs = args.episodes
    if "train" in sd:
        return int(sd["train"]["epoch"])
    done = int(sd["stats_i"][1])                      # Q2048_ST_EPISODES: files written before "train" existed
    return min(args.episodes, done // max(total_envs, 1))


def _load(path, rank):
    import torch

    return torch.load(path if rank == 0 and not os.path.exists(f"{path}.rank0") else f"{path}.rank{rank}",
                      map_location="cpu", weights_only=False)


def _save(agent, path, world, rank):
    """One file per rank (every rank trains its own replica): PATH for a single process, PATH.rankR in a job."""
    import torch

    batched = agent._b if hasattr(agent, "_b") else agent
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    sd = batched.state_dict(compact=True)
    sd["train"] = dict(getattr(batched, "train_progress", {"epoch": 0}))     # epochs finished (resume)
    env = getattr(batched, "train_env", None)
    if env is not None:
        sd["env"] = env.state_dict()                                         # the games in progress
    # (protocol 4: the rows of a long run are numpy arrays of more than 4 GiB -- 5.10^8 rows are 4 GB of keys
    # and 8 GB of values -- which the default protocol of torch.save refuses)
    torch.save(sd, path if world == 1 else f"{path}.rank{rank}", pickle_protocol=4)


def main(argv=None):
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the parent of the job has made no GPU call: one fresh process per rank (launch.py)
        spec = importlib.util.spec_from_file_location(
            "q2048_launch", os.path.join(REPO, "2048_q-learning_amd", "launch.py"))
        launcher = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(launcher)
        cmd = [sys.executable, os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv)
        raise SystemExit(launcher.launch_ranks(cmd, args.gpus, timeout=args.launch_timeout))
    pkg = importlib.import_module("2048_q-learning_amd")
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    single = args.num_envs == 1 and world == 1
    if args.summary and not single and not args.episode_log:
        raise SystemExit("--summary needs a per-episode log: --episode-log in batched mode")
    agent = train_single(args, pkg) if single else train_batched(args, pkg)
    if args.save:
        _save(agent, args.save, world, rank)
    if args.summary:                               # one row per log, like the reference's aggregate
        episodes_csv = args.log if single else (args.episode_log if world == 1 else f"{args.episode_log}.rank{rank}")
        out = args.summary if world == 1 else f"{args.summary}.rank{rank}"
        pkg.write_summary([pkg.summarize_csv(episodes_csv)], out)
    return agent


if __name__ == "__main__":
    main()
