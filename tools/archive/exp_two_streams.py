#!/usr/bin/env python3
"""Measurement tool: the batched 4-call loop (choose_action -> step ->
update_q_value -> reset(done)) with the batch as P independent env / agent pairs, each on a stream of
its own and with NO events between them -- every stream runs its own loop -- on ONE shared Q-table.
In a one-step launch every wave reads, then claims, then writes (DESIGN 4 "the 4-call surface"); two
streams put two such launches on the device at different phases.  The GPU side alone is timed: all
launches are queued behind a blocker kernel first, so the host's launch rate (8 launches per step with
two pairs, from one Python thread) is not what is measured.

    python tools/archive/exp_two_streams.py            # P = 1, 2, 4 at 1 Mi boards, alternating, 3 rounds
"""
import importlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("2048_q-learning_amd")
dev = torch.device("cuda:0")
B_TOTAL, STEPS, CAP = 1 << 20, int(os.environ.get("TWO_STREAMS_STEPS", "48")), 30


def run(P: int, table_from=None):
    B = B_TOTAL // P
    streams = [torch.cuda.Stream(dev) for _ in range(P)]
    envs, agents = [], []
    for k in range(P):
        env = pkg.BatchedGame2048Env(B, seed=0, env_id0=k * B, device=dev)
        agent = pkg.BatchedQLearningAgent(1000, learning_rate=0.1, discount_factor=0.99, exploration_rate=0.95,
                                          capacity_log2=CAP if k == 0 else 4, seed=0, env_id0=k * B, device=dev,
                                          placement="plain" if k else "auto")
        if k:                                            # one shared table: the same job as P = 1
            agent.table, agent.capacity_log2 = agents[0].table, agents[0].capacity_log2
        agent.fused_rollout(env, 256, play_only=True)    # mid-game boards
        agent.ctr = env.ctr
        envs.append(env); agents.append(agent)
    torch.cuda.synchronize()
    blocker_env = pkg.BatchedGame2048Env(1 << 20, seed=9, device=dev)
    blocker = pkg.BatchedQLearningAgent(10, exploration_rate=1.0, capacity_log2=4, seed=9, device=dev, placement="plain")

    def loops(steps):
        states = [e.boards for e in envs]
        for _ in range(steps):
            for k in range(P):
                with torch.cuda.stream(streams[k]):
                    a = agents[k].choose_action(states[k])
                    s2, r, d, _ = envs[k].step(a)
                    agents[k].update_q_value(states[k], a, r, s2, d)
                    states[k] = envs[k].reset(d)

    loops(4)
    torch.cuda.synchronize()
    # the blocker: ~25 ms of learner-less play on stream 0; every stream starts its loop behind it
    with torch.cuda.stream(streams[0]):
        blocker.fused_rollout(blocker_env, 1400, play_only=True)
        go = torch.cuda.Event(enable_timing=True)
        go.record(streams[0])
    for s in streams[1:]:
        s.wait_event(go)
    loops(STEPS)
    ends = []
    for s in streams:
        e = torch.cuda.Event(enable_timing=True)
        e.record(s)
        ends.append(e)
    queued_before_go = not go.query()                    # the whole loop was queued while the blocker ran
    torch.cuda.synchronize()
    ms = max(go.elapsed_time(e) for e in ends)
    rows = agents[0].table_size()
    inserts = sum(a.stats()["inserts"] for a in agents)
    out = {"pairs": P, "boards": B_TOTAL, "steps": STEPS, "us_per_step": round(ms * 1e3 / STEPS, 2),
           "env_steps_per_s": B_TOTAL * STEPS / ms * 1e3, "queued_behind_blocker": bool(queued_before_go),
           "rows": rows, "inserts": inserts, "cap_log2": CAP}
    print(json.dumps(out), flush=True)
    del envs, agents, blocker, blocker_env
    torch.cuda.empty_cache()


for rnd in range(int(os.environ.get("TWO_STREAMS_ROUNDS", "3"))):
    for P in (1, 2, 4):
        run(P)
