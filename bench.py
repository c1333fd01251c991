#!/usr/bin/env python3
"""bench.py -- env-steps/s of the fused 2048 Q-learning step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (choose -> env.step -> TD update -> reset-on-done,
Agent/main.py:91-101) over the whole batch: 1,048,576 boards per GPU (BASELINE configs[2] at
N = 1; configs[3] = 8 x 1,048,576 at N = 8, weak scaling).

With N > 1 and no process group in the environment the script starts its own N ranks (one fresh
child process per GPU, before anything in the parent touches the GPU) and relays rank 0's line;
under `python -m torch.distributed.run --nproc-per-node N` it joins the group it is given.

Protocol, per rank:
  1. input synthesis, untimed: `--prep-steps` steps of uniformly random play with no learner
     (Q2048_FLAG_PLAY_ONLY, eps = 1), so that boards and episode phases are the stationary
     mid-game mix and not 1 M boards five moves from reset;
  2. W warm-up steps of the real loop (table warm, RCCL initialised), untimed;
  3. the K-step region, timed `--repeats` times back to back on the same run (each region =
     ceil(K / steps_per_launch) launches + the statistics of the region on the host, bracketed by
     barrier + synchronize -- with several ranks the closing barrier is the statistics reduction, one
     collective over all ranks that nobody passes before every rank's steps are in; with one process
     the statistics are read from the rollout's host-side mirror after the closing synchronize; MAX
     over ranks): the MEDIAN region is the one reported.  Every region must
     finish episodes (`stats.episodes > 0`), or the reset / terminal-row path was not measured.
Boards, aux records and the hash Q-table are resident in HBM throughout.  Rank 0 prints ONE
JSON line.  The launches go through q2048_fused_rollout_opts: every env's row passes from launch to
launch through the row cache (`--no-row-cache`: each launch starts from a probe instead), and with one
process the region's statistics are read from the host-side mirror the launch's last block writes
(`--stats-by-copy`: a device-to-host copy queued behind the launches, as with several ranks).

Extra objects on the line:
  roofline      HBM roofline of the dominant kernel (k_fused_rollout): algorithmic bytes per
                launch (122 B per env-step) / average launch duration measured
                with HIP events on the launching stream, against the 8 TB/s HBM3E peak; `traffic`
                (and `fabric_frac` = traffic / launch time / peak: how busy the memory system is, next to
                `frac`, how well it is used, and `physical_minimum`, the run's own useful bytes) only when
                the committed PMC passes were taken with this run's configuration on these kernel sources.
  cpu_baseline  the CPU oracle (a C port of the reference loop, oracle/; kind "port") timed on this host's
                cores on a bounded sample of the same workload (rank 0, after the GPU regions, at
                every N): all cores and one thread, with the CPU model; and, under `product_core`, the
                product's own CPU twin (libq2048_host.so, device "cpu"; kind "product-core") on the same
                cores: the kernels' per-lane arithmetic and table, one shared table as on the GPU.
  companions    the same protocol at SURVEY 8(d)'s 2^28-slot table, at eps = 0.01, on 5x5 boards and on a FROZEN
                table -- pre-filled (untimed) to freeze_load 0.5, key set closed (Q2048_FLAG_NO_NEW_ROWS): where a
                run lives once its table has reached the largest capacity the device holds, i.e. after the first
                ~4e9 env-steps per GPU; the main line's table is young (N = 1 only).

`--check-shards` is a different job (no timing): every rank plays K steps of its shard at eps = 1
and rank 0 prints 64-bit hashes of boards + aux per 4096 global env ids -- equal lists for N = 1
and N = 2 over the same global ids prove, across processes, that a trajectory does not depend on
how the batch is sharded (tests/test_gpu_parity.py::test_check_shards_across_processes).
"""
from __future__ import annotations

import argparse
import hashlib
import importlib
import importlib.util
import json
import os
import statistics
import sys
import time

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
FREEZE_LOAD = 0.5           # the frozen companion: BatchedQLearningAgent's default freeze_load
PMC_TRAFFIC_FILES = [os.path.join(REPO, "profiles", f) for f in ("r06_pmc_traffic.json", "r06_pmc_traffic_k20.json",
                                                                 "r06_pmc_traffic_frozen_k20.json")]
CSRC = os.path.join(REPO, "2048_q-learning_amd", "csrc")
KERNEL_SOURCES = ("q2048_kernels.hip", "q2048_core.hpp", "q2048_core5.hpp", "q2048_luts.inc")


def kernel_sources_sha16() -> str:
    """Identifies the kernels a counter profile was taken on: SHA-256 over the HIP sources."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(CSRC, name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=256)
    p.add_argument("--warmup", type=int, default=64)
    p.add_argument("--repeats", type=int, default=5, help="timed K-step regions; the median is reported")
    p.add_argument("--prep-steps", type=int, default=1024,
                   help="input synthesis: untimed steps of random play without a learner (>= 64)")
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
    p.add_argument("--prefill-load", type=float, default=0.0,
                   help="measurement runs: fill the table, untimed, with this share of rows that are no states before the "
                        "run (>= the agent's freeze_load 0.5: the key set closes at the first launch -- the frozen "
                        "companion's workload as the main line, for the counter passes)")
    p.add_argument("--placement", default="auto",
                   type=lambda v: v if v in ("auto", "plain", "chunks") else int(v),
                   help="table allocation (agent.place_table): auto | chunks (2 MiB physical chunks, "
                        "q2048_table_alloc) | plain (hipMalloc) | N (best of N probed candidates)")
    p.add_argument("--strict-td", action="store_true",
                   help="TD write by compare-and-swap loop (Q2048_FLAG_TD_CAS) instead of one store; bounded: "
                        "after 16 lost races an update is stored plainly and counted (stats.cas_fallbacks)")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget; 0 = skip")
    p.add_argument("--experiment-bits", type=lambda v: int(v, 0), default=0,
                   help="unstable tuning bits OR-ed into the fused kernel's flags (ablations; not ABI)")
    p.add_argument("--no-row-cache", action="store_true",
                   help="A/B: launches start from a probe of the table instead of the row cache their "
                        "predecessor left (q2048_rollout_opts.row_cache)")
    p.add_argument("--stats-by-copy", action="store_true",
                   help="A/B: a region's statistics come back by a device-to-host copy queued behind the "
                        "launches (StatsAllReduce) instead of the rollout's host-side mirror")
    p.add_argument("--no-companions", action="store_true",
                   help="skip the 2^28-slot, eps = 0.01 and 5x5 companion runs (N = 1 only anyway)")
    p.add_argument("--check-shards", action="store_true",
                   help="no timing: play --steps steps at eps = 1 and print per-chunk hashes of boards + aux "
                        "(compare N = 1 with N = 2 over the same global env ids)")
    p.add_argument("--device", default="cuda",
                   help='--check-shards only: "cpu" plays the shards on the CPU twin (libq2048_host.so) over gloo -- the '
                        "N-rank partition rehearsed without GPUs (the timed bench is the MI355X path and takes no device)")
    p.add_argument("--launch-timeout", type=float, default=3000.0,
                   help="self-launched ranks (--gpus N > 1 outside torchrun) are stopped after this many seconds")
    return p.parse_args(argv)


def load_launcher():
    """The rank launcher, loaded by path so that the parent imports neither torch nor the package."""
    spec = importlib.util.spec_from_file_location(
        "q2048_launch", os.path.join(REPO, "2048_q-learning_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def committed_pmc_traffic(cfg: dict):
    """(bytes per env-step, source, None) when one of the committed rocprofv3 PMC profiles
    (profiles/r06_pmc_traffic*.json: the SURVEY protocol's 64-step launches, the driver's single
    20-step launch) was taken with this run's configuration AND on these kernel sources
    (`kernel_sources_sha16`, written by tools/pmc_summary.py on the box that ran the passes), else
    (None, None, what the committed passes were taken with)."""
    seen = []
    sha = kernel_sources_sha16()
    for path in PMC_TRAFFIC_FILES:
        try:
            with open(path) as fh:
                pmc = json.load(fh)
        except (OSError, ValueError):
            continue
        have = dict(pmc.get("config", {}), kernel_sources_sha16=pmc.get("kernel_sources_sha16"))
        if all(have.get(k) == v for k, v in cfg.items()) and have["kernel_sources_sha16"] == sha:
            return pmc["bytes_per_env_step"], pmc["source"], None
        seen.append(have)
    return None, None, (seen[0] if len(seen) == 1 else seen) if seen else None


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, seconds: float) -> dict:
    """Times the CPU oracle (kind 'port') on a bounded sample of the same workload: one thread,
    16, 32, a quarter of the cores and all usable cores, each with a private Q-table per thread; the
    best is reported, all are listed.  (More threads is not more throughput here: on the GPU box's
    2 x 64-core host the total peaks at 16-32 threads -- random accesses into tens of MiB of table per
    thread -- and falls to less than half on all 256 hardware threads, profiles/r03_cpu_mt.jsonl.)"""
    from oracle import oracle as O

    O.lib()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    runs = {}
    per = max(seconds / 5.0, 0.2)
    for T in sorted({1, min(16, cores), min(32, cores), max(1, cores // 4), cores}):
        B, steps = 16384 * T, 24
        envs = O.envs_init(B, 4, args.seed, 0)
        agents = [O.Agent(1000, 4, args.alpha, args.gamma, args.eps) for _ in range(T)]
        for a in agents:
            a.reserve(16384 * 64)      # rows one thread can create in this sample (no rehash)
        O.rollout_mt(envs, agents, 8, args.seed, 0, 0)                 # warm-up
        done, t0 = 0, time.perf_counter()
        budget = per
        ctr = 8
        while True:
            O.rollout_mt(envs, agents, steps, args.seed, 0, ctr)
            ctr += steps
            done += B * steps
            dt = time.perf_counter() - t0
            if dt > budget or ctr > 56:
                break
        runs[T] = (done / dt, B, ctr - 8, dt)
        del agents, envs
    T = max(runs, key=lambda t: runs[t][0])
    rate, B, steps, dt = runs[T]
    return {"value": rate, "unit": "env-steps/s", "cores": T, "kind": "port",
            "sample": f"oracle/q2048_oracle.c orc_rollout_mt: {B} boards x {steps} steps, "
                      f"{T} thread(s) with one private Q-table each, {dt:.1f} s "
                      f"(host has {cores} usable cores)",
            "single_thread": {"value": runs[1][0], "cores": 1,
                              "sample": f"{runs[1][1]} boards x {runs[1][2]} steps, {runs[1][3]:.1f} s"},
            "by_threads": {str(t): r[0] for t, r in sorted(runs.items())},
            "cpu_model": cpu_model(), "usable_cores": cores,
            "reference_python_1core_survey_container": 11144.0}


def prefill(torch, agent, load: float, seed: int) -> int:
    """Untimed: pseudo-random rows (keys that are no state of any game in this run, zero values) imported until
    the table holds `load` x its capacity -- the table of a run that has been going on for a while, where a
    growing table (capacity_log2="auto") spends its life: between a quarter of its load limit and the limit."""
    gen = torch.Generator(device=agent.device)
    gen.manual_seed(0x2048 + seed)
    want, chunk = int(load * (1 << agent.capacity_log2)), 1 << 25
    zeros = torch.zeros((chunk, 4), dtype=torch.float32, device=agent.device)
    words = 1 if agent.board_size == 4 else 2
    while want > 0:
        n = min(chunk, want)
        keys = torch.randint(-(1 << 62), 1 << 62, (n, words), dtype=torch.int64, device=agent.device, generator=gen)
        keys |= (-(1 << 63)) if words == 2 else 1        # 5x5 key words carry bit 63; a 4x4 key is never 0
        agent.import_rows_device(keys.view(-1) if words == 1 else keys, zeros[:n])
        want -= n
    del zeros
    if agent.check_status() & 4:                         # Q2048_STATUS_TABLE_FULL
        raise SystemExit("prefill: the table dropped rows")
    return agent.recount_rows()


def cpu_baseline_product_core(pkg, torch, args, seconds: float) -> dict:
    """Kind 'product-core': the product's own CPU twin (libq2048_host.so: the C ABI compiled for the host from the
    kernels' per-lane arithmetic, csrc/q2048_core.hpp -- SWAR boards, counter RNG, the same hash table) on this
    host's cores, through the same Python classes with device="cpu": the same workload as the GPU line (one
    SHARED table, Hogwild writes), bounded.  What the host cores can do with the product's code, next to the
    reference's loop ported to C (kind 'port')."""
    import numpy as np

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    runs, keep = {}, os.environ.get("Q2048_HOST_THREADS")
    per = max(seconds / 5.0, 0.2)
    try:
        for T in sorted({1, min(16, cores), min(32, cores), max(1, cores // 4), cores}):
            os.environ["Q2048_HOST_THREADS"] = str(T)
            B, steps = min(16384 * T, 1 << 21), 24            # (at most 2 Mi boards: an 8 GiB host table)
            env = pkg.BatchedGame2048Env(B, seed=args.seed, device="cpu")
            cap = max(20, int(np.ceil(np.log2(2.0 * B * 64))))
            agent = pkg.BatchedQLearningAgent(1000, learning_rate=args.alpha, discount_factor=args.gamma,
                                              exploration_rate=args.eps, capacity_log2=cap, seed=args.seed, device="cpu")
            agent.fused_rollout(env, 8)                               # warm-up
            done, t0 = 0, time.perf_counter()
            while True:
                agent.fused_rollout(env, steps)
                done += B * steps
                dt = time.perf_counter() - t0
                if dt > per or env.ctr > 56:
                    break
            runs[T] = (done / dt, B, env.ctr - 8, dt)
            del agent, env
    finally:
        if keep is None:
            os.environ.pop("Q2048_HOST_THREADS", None)
        else:
            os.environ["Q2048_HOST_THREADS"] = keep
    T = max(runs, key=lambda t: runs[t][0])
    rate, B, steps, dt = runs[T]
    return {"value": rate, "unit": "env-steps/s", "cores": T, "kind": "product-core",
            "sample": f"libq2048_host.so q2048_fused_rollout: {B} boards x {steps} steps, {T} thread(s) on ONE shared "
                      f"table, {dt:.1f} s (host has {cores} usable cores)",
            "single_thread": {"value": runs[1][0], "cores": 1},
            "by_threads": {str(t): r[0] for t, r in sorted(runs.items())}}


def measure(pkg, torch, args, dev, shard, world, *, eps, cap_log2, placement, steps, warmup, repeats,
            S, reducer, board_size=None, prefill_load=0.0, expect_frozen=False):
    """The protocol of the module docstring for one configuration.  Returns a dict of raw
    measurements (region times are MAX over ranks)."""
    board_size = args.board_size if board_size is None else board_size
    env = pkg.BatchedGame2048Env(shard.num_envs, board_size=board_size, seed=args.seed,
                                 env_id0=shard.env_id0, device=dev)
    if args.agent == "row-tuple":
        agent = pkg.BatchedRowTupleAgent(1000, learning_rate=args.alpha, discount_factor=args.gamma,
                                         exploration_rate=eps, seed=args.seed,
                                         env_id0=shard.env_id0, device=dev)
        synth = pkg.BatchedQLearningAgent(1000, exploration_rate=1.0, capacity_log2=4, seed=args.seed,
                                          env_id0=shard.env_id0, device=dev, board_size=4,
                                          placement="plain")
    else:
        agent = pkg.BatchedQLearningAgent(1000, learning_rate=args.alpha, discount_factor=args.gamma,
                                          exploration_rate=eps, capacity_log2=cap_log2,
                                          seed=args.seed, env_id0=shard.env_id0, device=dev,
                                          strict_td=args.strict_td, board_size=board_size,
                                          placement=placement, row_cache=not args.no_row_cache)
        synth = agent
        agent.experiment_bits = args.experiment_bits
    prefilled_rows = prefill(torch, agent, prefill_load, args.seed) if prefill_load > 0 else 0

    def run(steps_):
        launches, left = 0, steps_
        while left > 0:
            k = min(S, left)
            agent.fused_rollout(env, k)
            left -= k
            launches += 1
        return launches

    # 1. input synthesis: random play, no learner, nothing of the table touched
    keep_eps, synth.epsilon = synth.epsilon, 1.0
    left = args.prep_steps
    while left > 0:
        k = min(256, left)
        synth.fused_rollout(env, k, play_only=True)
        left -= k
    synth.epsilon = keep_eps
    synth.ctr = agent.ctr = env.ctr
    synth.stats(reset=True)
    prep_episodes_per_env = float(env.aux_fields()["episode"].mean())

    # 2. warm-up of the real loop
    run(warmup)
    reducer.start(agent.stats_i, agent.stats_f)      # untimed: RCCL builds its rings lazily
    reducer.wait()

    # 3. timed regions.  What a region holds besides its launches is kept small on purpose: the two HIP
    # events exist before the clock starts; with one process the statistics come back through the
    # rollout's own host-side mirror (written by the launch's last block: nothing is queued behind the
    # kernel) and the closing synchronize is the only wait; with several ranks the statistics reduction
    # -- ONE collective, on its own stream -- is the closing barrier: nobody gets past it before every
    # rank's K steps are in.  MAX over ranks of every region's times: one collective after the last region.
    mirrored = (not torch.distributed.is_initialized() and hasattr(agent, "mirrored_stats")
                and not args.stats_by_copy)
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(repeats)]
    regions = []
    for ev0, ev1 in events:
        agent.stats(reset=True)
        torch.cuda.synchronize(dev)
        pkg.dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        ev0.record()
        launches = run(steps)                        # exactly K steps
        ev1.record()
        if mirrored:
            torch.cuda.synchronize(dev)
            si, sf = agent.mirrored_stats()
        else:
            reducer.start(agent.stats_i, agent.stats_f, snapshot=False)   # the path's only collective; nothing
            # else touches the vectors before the region's closing wait: read in place, no clone, no side stream
            si, sf = reducer.wait()
            torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t0
        st = pkg.stats_dict(si, sf)                  # all-reduced: whole-job numbers
        total = shard.total_envs * steps
        assert st["steps"] == total, (st["steps"], total)
        assert st["episodes"] > 0, "no episode finished inside a timed region: reset path unmeasured"
        regions.append({"wall_s": wall, "kernel_ms": ev0.elapsed_time(ev1), "launches": launches, "stats": st})
    if mirrored:                                     # the mirror against the device vectors themselves
        mi, mf = agent.mirrored_stats()
        assert (mi == agent.stats_i.cpu().numpy()).all() and (mf == agent.stats_f.cpu().numpy()).all(), \
            "statistics mirror != device vectors"
    worst = pkg.dist.max_over_ranks_many([r["wall_s"] for r in regions] + [r["kernel_ms"] for r in regions],
                                         device=dev)
    for k, r in enumerate(regions):
        r["wall_s"], r["kernel_ms"] = worst[k], worst[len(regions) + k]
    table_rows = agent.table_size() if args.agent == "hash" else None
    status = agent.check_status()
    frozen = bool(getattr(agent, "frozen", False))
    if expect_frozen and (not frozen or table_rows != prefilled_rows):
        raise SystemExit(f"the frozen companion's table was not frozen ({frozen}) or took rows ({table_rows} != {prefilled_rows})")
    placement_report = getattr(agent, "placement", None)
    del agent, synth, env
    torch.cuda.empty_cache()
    return {"regions": regions, "table_rows": table_rows, "status": status, "mirrored": mirrored,
            "placement": placement_report, "prep_episodes_per_env": prep_episodes_per_env,
            "prefilled_rows": prefilled_rows, "frozen": frozen}


def summarise(m, shard, steps, algo_bytes):
    """Median region -> value, ms_per_step, roofline numbers."""
    walls = [r["wall_s"] for r in m["regions"]]
    med = statistics.median_low(walls)
    reg = m["regions"][walls.index(med)]
    launches = reg["launches"]
    avg_launch_s = sum(r["kernel_ms"] for r in m["regions"]) / 1e3 / (launches * len(m["regions"]))
    achieved = algo_bytes * shard.num_envs * (steps / launches) / avg_launch_s / 1e9
    return {"value": shard.total_envs * steps / med, "ms_per_step": med * 1e3 / steps,
            "region_ms": [round(w * 1e3, 4) for w in walls], "median_region": reg,
            "avg_launch_s": avg_launch_s, "launches": launches, "achieved_gbs": achieved}


def run_rank(args):
    import torch

    # a rank of a multi-GPU job sits on the CPUs next to its GPU (sysfs + sched_setaffinity, before any GPU call);
    # the CPU leg at the end gets the process's original CPUs back
    cpus_before = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    pinned = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # (the device this rank will really use: one GPU per local rank, ranks sharing GPUs only in rehearsals on a
        # smaller box; torch.cuda.device_count() does not initialise the GPU)
        pinned = load_launcher().pin_to_gpu_numa_node(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
    pkg = importlib.import_module("2048_q-learning_amd")
    if args.experiment_bits:          # ablation bits exist in the measurement build only
        pkg._native.use_experiments_build()
    rank, local_rank, world = pkg.dist.init_process_group()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible")
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())  # 1 rank = 1 GPU on a node
    torch.cuda.set_device(dev)

    B = args.boards_per_gpu
    shard = pkg.weak_shard(B, world, rank)
    S = max(1, min(args.steps_per_launch, args.steps))
    learn_steps = args.warmup + args.repeats * args.steps
    # every step may create a row: load <= 0.5 at the end of the run, and at least 2^30 slots (32 GiB,
    # mapped from 2 MiB chunks: as fast as a table spanning 128 GiB, DESIGN.md 4 "table placement")
    cap_log2 = args.cap_log2 or pkg.auto_capacity_log2(B * max(learn_steps, 1), dev, max_log2=32)
    if args.agent == "hash" and args.cap_log2:
        # a table given by hand must still end the run below load ~0.6 (0.75 new rows per board-step
        # at the most), or every region is slower than the one before it: fewer regions, then
        budget = int(0.6 * (1 << cap_log2) / (0.75 * B))
        while args.repeats > 1 and args.warmup + args.repeats * args.steps > budget:
            args.repeats -= 1
        if args.warmup + args.repeats * args.steps > budget:
            raise SystemExit(f"--cap-log2 {cap_log2} cannot hold {args.warmup} + {args.steps} steps of "
                             f"{B} boards below load 0.6")
    algo_bytes = ALGO_BYTES_FUSED_4X4 if args.board_size == 4 else ALGO_BYTES_FUSED_5X5
    if args.agent == "row-tuple":
        algo_bytes = ALGO_BYTES_ROW_TUPLE
    reducer = pkg.StatsAllReduce(dev)

    m = measure(pkg, torch, args, dev, shard, world, eps=args.eps, cap_log2=cap_log2,
                placement=args.placement, steps=args.steps, warmup=args.warmup,
                repeats=args.repeats, S=S, reducer=reducer, prefill_load=args.prefill_load,
                expect_frozen=args.prefill_load >= FREEZE_LOAD)
    s = summarise(m, shard, args.steps, algo_bytes)
    st = s["median_region"]["stats"]

    kernel = "k_rt_fused_rollout" if args.agent == "row-tuple" else "k_fused_rollout"
    roofline = {"bound": "hbm", "kernel": kernel, "achieved": s["achieved_gbs"],
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": s["achieved_gbs"] / HBM_PEAK_GBS,
                "frac_of_measured_copy_ceiling": s["achieved_gbs"] / HBM_COPY_CEILING_GBS,
                "traffic": None,
                "algorithmic_bytes_per_env_step": algo_bytes,
                "avg_launch_ms": s["avg_launch_s"] * 1e3, "launches": s["launches"],
                "note": f"register-resident, K={S} env steps per launch: boards/aux cross HBM once "
                        f"per launch, the figure counts them once per step (SURVEY 8(d))"}
    if args.agent == "hash":
        # what a step has to move at the least, from this run's own statistics: the board + aux
        # image once per launch, one 32-B row read for every move that reaches another state, the
        # 4-B Q write, the 8-B key of every new row
        moved = st["valid_moves"] / max(st["steps"], 1)
        new_rows = st["inserts"] / max(st["steps"], 1)
        cells = args.board_size ** 2
        phys = (2 * (cells + 16)) / S + 32.0 * moved + 4.0 + (8.0 if args.board_size == 4 else 16.0) * new_rows
        per_launch_phys = phys * shard.num_envs * (args.steps / s["launches"])
        roofline["physical_minimum"] = {
            "bytes_per_env_step": phys, "achieved": per_launch_phys / s["avg_launch_s"] / 1e9,
            "frac": per_launch_phys / s["avg_launch_s"] / 1e9 / HBM_PEAK_GBS,
            "note": "useful bytes only; what the memory system moves for them is `traffic`: every L2 "
                    "miss is a 128-B read request whatever the load's width or cache policy, write-backs "
                    "are 32 B, atomics 64 B (profiles/r02_requests/)"}
        cfg = {"boards": shard.num_envs, "steps_per_launch": S, "cap_log2": cap_log2,
               "board_size": args.board_size, "eps": args.eps, "strict_td": bool(args.strict_td)}
        if args.prefill_load:
            cfg["prefill_load"] = args.prefill_load
        per_step, source, other = committed_pmc_traffic(cfg)
        if per_step is not None:
            roofline["traffic"] = per_step * shard.num_envs * (args.steps / s["launches"])
            roofline["traffic_bytes_per_env_step"] = per_step
            roofline["traffic_source"] = source
            # what the memory fabric actually carried per second of this launch, against the peak: `frac` prices
            # the algorithm's bytes, this one the requests the kernel really issued (128-B read granules, 32-B
            # write-backs, 64-B atomics) -- how busy the memory system is, as opposed to how well it is used
            roofline["fabric_frac"] = roofline["traffic"] / s["avg_launch_s"] / 1e9 / HBM_PEAK_GBS
        elif other is not None:
            roofline["traffic_profile_config"] = other   # the committed passes are for another run

    out = {
        "metric": "env_steps_per_sec", "value": s["value"], "unit": "env-steps/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": s["ms_per_step"],
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
                   "table_placement": m["placement"], "experiment_bits": args.experiment_bits,
                   "row_cache": not args.no_row_cache, "prefill_load": args.prefill_load, "frozen": m["frozen"],
                   "region_statistics": "host-side mirror written by the launch's last block" if m["mirrored"] else (
                       "all-gather (one collective)" if torch.distributed.is_initialized() else "copy to pinned host memory"),
                   "prep_steps": args.prep_steps, "repeats": args.repeats,
                   "timing": "median of `repeats` K-step regions after `prep_steps` of random play "
                             "and `warmup` learning steps",
                   "parallelism": f"env-batch x{world}, RCCL all-reduce of statistics only "
                                  f"(side stream)"},
        "region_ms": s["region_ms"],
        "roofline": roofline,
        "stats": {"episodes": st["episodes"], "mean_return": st["mean_return"],
                  "mean_score": st["mean_score"], "valid_move_frac": st["valid_moves"] / max(st["steps"], 1),
                  "table_inserts": st["inserts"], "inserts_per_step": st["inserts"] / max(st["steps"], 1),
                  "drops": st["drops"],
                  "table_rows_per_gpu": m["table_rows"], "claim_timeouts": pkg._native.claim_timeouts(),
                  "table_load_factor": None if m["table_rows"] is None else m["table_rows"] / float(1 << cap_log2),
                  "cas_retries": st["cas_retries"], "cas_fallbacks": st["cas_fallbacks"], "status": m["status"],
                  "episodes_per_env_before_timing": m["prep_episodes_per_env"],
                  "max_tile_hist": {str(k): v for k, v in st["max_tile_hist"].items()}},
        "kernel_ms_total": sum(r["kernel_ms"] for r in m["regions"]),
    }

    if world == 1 and args.agent == "hash" and not args.no_companions:
        # the 2^28-slot table the survey specified, the exploit-heavy eps = 0.01 run,
        # and BASELINE configs[4] (5x5 boards, 156 algorithmic bytes per env-step); each sized so
        # that its table ends below load 0.5
        comps = []
        # (rounds 2-3 gave the 5x5 companion a 2^32-slot table: 0.34 against 0.30 on 32 GiB then.  Alternating
        # runs on one box, round 4: 63.0 / 65.4 / 65.2 us per step on 2^30 slots in chunks against 65.0 / 64.5 /
        # 64.9 on 2^32 -- profiles/r04_5x5_capacity_ab.txt -- so it runs on the main line's table size)
        cap5 = cap_log2
        for name, c_eps, c_cap, c_n, c_fill in (
                ("capacity 2^28 (SURVEY 8(d))", args.eps, 28, args.board_size, 0.0),
                ("epsilon 0.01 (argmax path)", 0.01, cap_log2, args.board_size, 0.0),
                ("5x5 boards (BASELINE configs[4])", args.eps, cap5, 5, 0.0),
                # the main line runs on a YOUNG table (load < 0.1 at the driver's K = 20): this is the same workload
                # on the table of a LONG run -- one that has reached its largest capacity and closed its key set at
                # freeze_load (pre-filled, untimed, to just above it: the agent's own policy freezes it at its first
                # launch): existing rows learn, absent states read as zeros, their updates are dropped and counted
                (f"frozen table: pre-filled to load {FREEZE_LOAD}, key set closed (Q2048_FLAG_NO_NEW_ROWS)", args.eps,
                 cap_log2, args.board_size, FREEZE_LOAD + 0.002)):
            if c_n == args.board_size and name.startswith("5x5"):
                continue
            # learning steps the table can take: up to load 0.5 (the frozen one takes no rows: no limit)
            budget = int(0.5 * (1 << c_cap) / (0.75 * B)) if not c_fill else args.warmup + 4 * args.steps
            c_rep = 3
            c_steps = max(1, min(args.steps, budget // (c_rep + 1)))
            c_warm = max(1, min(args.warmup, budget - c_rep * c_steps))
            c_S = max(1, min(S, c_steps))
            c_bytes = ALGO_BYTES_FUSED_4X4 if c_n == 4 else ALGO_BYTES_FUSED_5X5
            cm = measure(pkg, torch, args, dev, shard, world, eps=c_eps, cap_log2=c_cap,
                         placement=args.placement, steps=c_steps, warmup=c_warm, repeats=c_rep, S=c_S,
                         reducer=reducer, board_size=c_n, prefill_load=c_fill, expect_frozen=bool(c_fill))
            cs = summarise(cm, shard, c_steps, c_bytes)
            cst = cs["median_region"]["stats"]
            comps.append({"name": name, "epsilon": c_eps, "table_capacity_log2": c_cap, "board_size": c_n,
                          "steps": c_steps, "warmup": c_warm, "repeats": c_rep,
                          "steps_per_launch": c_S, "value": cs["value"],
                          "ms_per_step": cs["ms_per_step"], "region_ms": cs["region_ms"],
                          "algorithmic_bytes_per_env_step": c_bytes,
                          "roofline_frac": cs["achieved_gbs"] / HBM_PEAK_GBS,
                          "inserts_per_step": cst["inserts"] / max(cst["steps"], 1),
                          "episodes": cst["episodes"], "table_placement": cm["placement"],
                          "prefilled_rows": cm["prefilled_rows"], "frozen": cm["frozen"],
                          "drops_per_step": cst["drops"] / max(cst["steps"], 1),
                          "table_load_factor": cm["table_rows"] / float(1 << c_cap)})
            if c_fill:
                # the frozen workload has counter passes of its own (bench.py --prefill-load, tools/sessions/r06_final_a.sh)
                c_cfg = {"boards": shard.num_envs, "steps_per_launch": c_S, "cap_log2": c_cap, "board_size": c_n,
                         "eps": c_eps, "strict_td": bool(args.strict_td), "prefill_load": c_fill}
                c_per_step, c_source, _ = committed_pmc_traffic(c_cfg)
                if c_per_step is not None:
                    comps[-1]["traffic_bytes_per_env_step"] = c_per_step
                    comps[-1]["fabric_frac"] = c_per_step * shard.num_envs * c_steps / (cs["avg_launch_s"] * cs["launches"]) / 1e9 / HBM_PEAK_GBS
                    comps[-1]["traffic_source"] = c_source
        out["companions"] = comps

    # the GPU side is done: leave the process group before the host-only leg, so that no rank
    # waits in a collective while rank 0 times the CPU
    if torch.distributed.is_initialized():
        pkg.dist.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        if pinned is not None:
            out["config"]["rank0_cpu_affinity"] = {"numa_node": pinned[0], "cpus": len(pinned[1])}
            os.sched_setaffinity(0, cpus_before)
        # the reference's CPU path next to the GPU number, on this node's cores, at every N
        out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds) if args.cpu_seconds > 0 else None
        if out["cpu_baseline"] is not None:
            # ... and what the same cores do with the product's own code (the CPU twin): both figures on the line
            out["cpu_baseline"]["product_core"] = cpu_baseline_product_core(pkg, torch, args, args.cpu_seconds)
        print(json.dumps(out), flush=True)


SHARD_CHUNK = 4096


def check_shards(args):
    """`--check-shards`: every rank plays `--steps` steps of its contiguous shard of global env ids
    at eps = 1 (actions are draws: trajectories do not depend on the table) and hashes boards + aux
    per chunk of SHARD_CHUNK global ids; rank 0 prints all chunks in id order.  The same list for
    any world size over the same ids = a trajectory does not depend on the sharding."""
    import numpy as np
    import torch

    pkg = importlib.import_module("2048_q-learning_amd")
    on_cpu = args.device == "cpu"
    rank, local_rank, world = pkg.dist.init_process_group("gloo" if on_cpu else None)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if on_cpu:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: no HIP device visible")
        dev = torch.device("cuda", local_rank % torch.cuda.device_count())
        torch.cuda.set_device(dev)
    shard = pkg.weak_shard(args.boards_per_gpu, world, rank)
    if shard.num_envs % SHARD_CHUNK:
        raise SystemExit(f"--boards-per-gpu must be a multiple of {SHARD_CHUNK}")
    env = pkg.BatchedGame2048Env(shard.num_envs, board_size=args.board_size, seed=args.seed,
                                 env_id0=shard.env_id0, device=dev)
    cap = args.cap_log2 or max(16, int(np.ceil(np.log2(2.0 * shard.num_envs * args.steps))))
    agent = pkg.BatchedQLearningAgent(1000, learning_rate=args.alpha, discount_factor=args.gamma,
                                      exploration_rate=1.0, capacity_log2=cap, seed=args.seed,
                                      env_id0=shard.env_id0, device=dev, board_size=args.board_size,
                                      placement="plain")
    left = args.steps
    while left > 0:
        k = min(args.steps_per_launch, left)
        agent.fused_rollout(env, k)
        left -= k
    cells = args.board_size ** 2
    b = torch.zeros((shard.num_envs, 32), dtype=torch.uint8, device=dev)
    b[:, :cells] = env.boards
    words = torch.cat([b.view(torch.int64), env.aux.view(torch.int64)], dim=1)          # [B, 6]
    consts = (0x9E3779B97F4A7C15, 0xC2B2AE3D27D4EB4F, 0x165667B19E3779F9, 0x27D4EB2F165667C5,
              0x85EBCA77C2B2AE63, 0xFF51AFD7ED558CCD)
    mult = torch.tensor([c - (1 << 64) if c >> 63 else c for c in consts], dtype=torch.int64, device=dev)
    ids = torch.arange(shard.num_envs, dtype=torch.int64, device=dev) + shard.env_id0
    h = (words * mult).sum(dim=1) ^ (ids * 0x2545F4914F6CDD1D)                            # wraps mod 2^64
    h = (h ^ (h >> 29)) * -0x61C8864680B583EB
    chunks = h.view(-1, SHARD_CHUNK).sum(dim=1).cpu().numpy().view(np.uint64)
    st = agent.stats()
    mine = {"rank": rank, "env_id0": shard.env_id0, "hashes": [f"{int(x):016x}" for x in chunks],
            "episodes": st["episodes"], "steps": st["steps"], "status": agent.check_status()}
    if world > 1:
        gathered = [None] * world
        torch.distributed.all_gather_object(gathered, mine)
        pkg.dist.barrier()
        torch.distributed.destroy_process_group()
    else:
        gathered = [mine]
    if rank == 0:
        gathered.sort(key=lambda g: g["env_id0"])
        print(json.dumps({"check_shards": True, "n_gpus": world, "total_envs": shard.total_envs,
                          "steps": args.steps, "chunk": SHARD_CHUNK, "board_size": args.board_size,
                          "seed": args.seed, "hashes": sum((g["hashes"] for g in gathered), []),
                          "episodes": sum(g["episodes"] for g in gathered),
                          "env_steps": sum(g["steps"] for g in gathered),
                          "status": max(g["status"] for g in gathered)}), flush=True)


def main(argv=None):
    args = parse_args(argv)
    if args.steps < 1 or args.warmup < 0 or args.repeats < 1 or args.gpus < 1:
        raise SystemExit("--steps, --repeats, --gpus must be >= 1 and --warmup >= 0")
    args.prep_steps = max(64, args.prep_steps)      # at least 64 steps of synthesis
    launcher = load_launcher()
    if args.gpus > 1 and not launcher.inside_a_launch():
        # the parent of the job: it has made no GPU call, starts one fresh process per rank,
        # relays rank 0's JSON line and exits with the ranks' worst code
        cmd = [sys.executable, os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv)
        raise SystemExit(launcher.launch_ranks(cmd, args.gpus, timeout=args.launch_timeout,
                                               line_filter=lambda ln: ln.lstrip().startswith("{")))
    if args.check_shards:
        check_shards(args)
    elif args.device != "cuda":
        raise SystemExit("--device applies to --check-shards; the timed bench is the MI355X path")
    else:
        run_rank(args)


if __name__ == "__main__":
    main()
