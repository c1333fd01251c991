#!/usr/bin/env python3
"""Confidence run (not part of the test suite): the independent-lane parity check of
tests/test_gpu_parity.py over more seeds, both geometries, every env profile, deferred writes on and
off, random launch splits, and -- for a third of the cases -- the 4-call API (choose_action, step into
the other board buffer, update_q_value with the row cache, reset(done)) and the deterministic step instead of the
fused rollout, switching between the three in mid-run -- and, for half of the cases, a key set that CLOSES at a random step
(Q2048_FLAG_NO_NEW_ROWS: the oracle agents' `freeze()`; the envs' visit rows cross every launch boundary and every
switch between the fused rollout and the 4-call API in the row cache).  Boards and aux bit-exact, every Q row within
rtol 1e-5, the closed key sets exactly the oracle's."""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("2048_q-learning_amd")
# FUZZ_PRODUCT=1: the SHIPPED library (its template instantiations, no experiment bits: deferred writes
# always on); default: the measurement build, whose run-time write modes take bits 8..23 of `flags`
# FUZZ_DEVICE=cpu: the CPU twin (libq2048_host.so; no measurement build of it exists, so PRODUCT is implied) --
# tests/test_host_twin.py runs a few trials of this in the CPU suite
dev = os.environ.get("FUZZ_DEVICE", "cuda:0")
PRODUCT = bool(os.environ.get("FUZZ_PRODUCT")) or dev == "cpu"
if not PRODUCT:
    pkg._native.use_experiments_build()
from oracle import oracle as O  # noqa: E402  (the checker)

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    n = int(rng.choice([4, 5]))
    B, steps = int(rng.integers(50, 260)), int(rng.integers(100, 500))
    seed, id0 = int(rng.integers(0, 2 ** 31)), int(rng.integers(0, 2 ** 40))
    eps, lr, gamma = float(rng.choice([0.0, 0.05, 0.3, 1.0])), float(rng.choice([0.1, 0.5])), float(rng.choice([0.9, 0.99]))
    profile = str(rng.choice(["shaped", "nopenalty"]))
    reset_shaping = bool(rng.integers(0, 2))
    bits = int(rng.choice([0, 1 << 14]))                       # deferred same-state writes on / off
    bits = 0 if PRODUCT else bits
    strict = bool(rng.integers(0, 2))
    flags = (O.ENV_DQN if profile == "nopenalty" else 0) | (O.ENV_RESET_SHAPING if reset_shaping else 0)
    freeze_at = int(rng.integers(0, steps)) if rng.integers(0, 2) else None     # the key set closes before this step
    envs = O.envs_init(B, n, seed, id0)
    agents = [O.Agent(100, 4, lr, gamma, eps, n=n) for _ in range(B)]
    drops = 0
    for i in range(B):
        if freeze_at is None:
            O.rollout(envs[i:i + 1], agents[i], steps, seed, id0 + i, 0, env_flags=flags)
        else:
            if freeze_at:
                O.rollout(envs[i:i + 1], agents[i], freeze_at, seed, id0 + i, 0, env_flags=flags)
            agents[i].freeze()
            O.rollout(envs[i:i + 1], agents[i], steps - freeze_at, seed, id0 + i, freeze_at, env_flags=flags)
            drops += agents[i].drops
    env = pkg.BatchedGame2048Env(B, board_size=n, seed=seed, env_id0=id0, device=dev, profile=profile,
                                 reset_shaping_state=reset_shaping)
    agent = pkg.BatchedQLearningAgent(100, learning_rate=lr, discount_factor=gamma, exploration_rate=eps,
                                      capacity_log2=19, seed=seed, env_id0=id0, device=dev, independent=True,
                                      board_size=n, strict_td=strict, freeze_load=None)
    agent.experiment_bits = bits
    four_call = bool(rng.integers(0, 3) == 0)
    left = steps
    while left > 0:
        k = int(min(left, rng.integers(1, 200)))
        if freeze_at is not None:
            done_steps = steps - left
            if done_steps == freeze_at:
                agent.frozen = True                      # (closed by hand at the oracle's step; the policy has its own tests)
            elif done_steps < freeze_at:
                k = min(k, freeze_at - done_steps)       # a launch boundary exactly where the key set closes
        if four_call and rng.integers(0, 3) == 0:
            # (lanes with private rows: the two-phase step IS the sequential one; its visit rows use the same records)
            agent.deterministic_rollout(env, k)
        elif four_call and rng.integers(0, 2):
            state = env.boards
            for _ in range(k):
                a = agent.choose_action(state)
                nxt, r, d, _ = env.step(a)
                agent.update_q_value(state, a, r, nxt, d)
                state = env.reset(d)
        else:
            agent.fused_rollout(env, k)
        left -= k
    assert np.array_equal(env.boards.cpu().numpy(), envs["board"][:, :n * n]), (trial, "boards")
    f = env.aux_fields()
    assert np.array_equal(f["score"], envs["score"]) and np.array_equal(f["episode"], envs["episode"]), (trial, "aux")
    worst = 0.0
    for i, oa in enumerate(agents):
        keys, vals = oa.dump()
        if len(keys) == 0:                               # (a key set closed at step 0: this agent has no row at all)
            continue
        got = agent.q_values(torch.from_numpy(keys).to(dev), env_id=id0 + i).cpu().numpy()
        assert np.allclose(got, vals, rtol=1e-5, atol=1e-6 * max(1.0, float(np.abs(vals).max()))), (trial, i)
        worst = max(worst, float(np.max(np.abs(got - vals) / (np.abs(vals) + 1e-1))))
    assert agent.table_size() == sum(len(a) for a in agents) and agent.check_status() == 0
    assert agent.stats()["drops"] == drops, (trial, "drops", agent.stats()["drops"], drops)
    cases += 1
    print(f"trial {trial}: {n}x{n} B={B} steps={steps} eps={eps} profile={profile} reset_shaping={reset_shaping} "
          f"strict={strict} bits={bits:#x} four_call={four_call} freeze_at={freeze_at} ({drops} drops): ok, worst relative Q "
          f"error {worst:.2e}", flush=True)
print(f"{cases} cases passed")
