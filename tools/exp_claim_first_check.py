#!/usr/bin/env python3
"""Round 5: the claim-first probe of the 5x5 rollout (experiment bit 15, measurement build) against the oracle:
300 envs with private rows, 200 steps, epsilon 0.3 -- boards bit-exact, every Q row within 1e-5, table size equal --
and a shared table at epsilon 1 (20 000 envs x 60 steps): the key set is the oracle dict's."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("2048_q-learning_amd")
pkg._native.use_experiments_build()
from oracle import oracle as O
dev, n = "cuda:0", 5
B, steps, seed, id0, eps = 300, 200, 9, 500, 0.3
env = pkg.BatchedGame2048Env(B, board_size=n, seed=seed, env_id0=id0, device=dev)
agent = pkg.BatchedQLearningAgent(100, learning_rate=0.1, discount_factor=0.95, exploration_rate=eps, capacity_log2=18,
                                  seed=seed, env_id0=id0, device=dev, independent=True, board_size=n)
agent.experiment_bits = 1 << 15
for k in (50, 1, 149):
    agent.fused_rollout(env, k)
envs = O.envs_init(B, n, seed, id0)
total = 0
for i in range(B):
    oa = O.Agent(100, 4, 0.1, 0.95, eps, n=n)
    O.rollout(envs[i:i + 1], oa, steps, seed, id0 + i, 0)
    keys, vals = oa.dump()
    total += len(oa)
    got = agent.q_values(torch.from_numpy(keys).to(dev), env_id=id0 + i).cpu().numpy()
    assert np.allclose(got, vals, rtol=1e-5, atol=1e-6), i
assert np.array_equal(env.boards.cpu().numpy(), envs["board"][:, :25])
st = agent.stats()
assert agent.table_size() == total == st["inserts"] and st["drops"] == 0 and pkg._native.claim_timeouts() == 0
B2 = 20000
env = pkg.BatchedGame2048Env(B2, board_size=n, seed=3, device=dev)
agent = pkg.BatchedQLearningAgent(100, exploration_rate=1.0, capacity_log2=22, seed=3, device=dev, board_size=n)
agent.experiment_bits = 1 << 15
agent.fused_rollout(env, 60)
envs = O.envs_init(B2, n, 3, 0)
oa = O.Agent(100, 4, 0.1, 0.9, 1.0, n=n)
O.rollout(envs, oa, 60, 3, 0, 0)
assert np.array_equal(env.boards.cpu().numpy(), envs["board"][:, :25])
st = agent.stats()
assert agent.table_size() == len(oa) == st["inserts"] and pkg._native.claim_timeouts() == 0, (agent.table_size(), len(oa), st["inserts"])
print("claim-first: private rows == oracle, shared-table key set == oracle dict (", len(oa), "rows ), no claim time-outs")
