#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ FROM THE REFERENCE ITSELF.

Runs only in the build container, where /root/reference is mounted; the reference never
travels, only the vectors (inputs + expected outputs) written here do.  The reference's
tabular path is imported unmodified:

  * `gymnasium` is not installed; the path needs it only as a base class and two attribute
    holders (Game2048_env.py:1,3,78,89-90), so a ten-line stub module is registered first.
  * The reference draws from the global numpy / `random` streams (Game2048_env.py:19-20,
    Agent/main.py:35-36).  To make "identical seeds" meaningful, this script patches those
    four calls to return the decisions of the build's counter RNG (oracle.draws), i.e. the
    reference is driven by injected draws and its outputs are recorded.

Usage:  python tests/golden/generate_golden.py        (writes *.npz next to this file)
"""
from __future__ import annotations

import importlib.util
import json
import os
import random as pyrandom
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = os.environ.get("Q2048_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
from oracle import oracle as O  # noqa: E402  (only for the RNG + draw->decision maps)


# --------------------------------------------------------------------------------------
# import the reference
# --------------------------------------------------------------------------------------
def import_reference():
    sys.dont_write_bytecode = True
    os.environ.setdefault("MPLBACKEND", "Agg")
    if "gymnasium" not in sys.modules:
        gym = types.ModuleType("gymnasium")

        class Env:  # base class only
            pass

        spaces = types.ModuleType("gymnasium.spaces")

        class Discrete:
            def __init__(self, n):
                self.n = n

        class Box:
            def __init__(self, low, high, shape=None, dtype=None):
                self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

        spaces.Discrete, spaces.Box = Discrete, Box
        gym.Env, gym.spaces = Env, spaces
        sys.modules["gymnasium"] = gym
        sys.modules["gymnasium.spaces"] = spaces
    base = os.path.join(REF, "QLearningBase")
    if base not in sys.path:
        sys.path.insert(0, base)
    import environment.Game2048_env as env_mod  # noqa

    spec = importlib.util.spec_from_file_location(
        "ref_agent_main", os.path.join(base, "Agent", "main.py"))
    agent_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(agent_mod)  # __main__ guard keeps the training loop off
    return env_mod, agent_mod


# --------------------------------------------------------------------------------------
# draw injection
# --------------------------------------------------------------------------------------
class Feed:
    """Replaces np.random.randint/random and random.random/randint while active."""

    def __init__(self):
        self.env_q: list[int] = []   # raw u32 draws for the env: pos, val, pos, val ...
        self.agent_x = (0, 0)        # raw u32 draws for the agent: eps, action
        self.trial = False           # inside is_game_over: feed dummies (result discarded)
        self.calls = {"np_randint": 0, "np_random": 0, "py_random": 0, "py_randint": 0}

    # numpy side (Game2048_env.py:19-20)
    def np_randint(self, low, high=None, *a, **k):
        self.calls["np_randint"] += 1
        n = high if high is not None else low
        if self.trial:
            return 0
        return O.draw_index(self.env_q.pop(0), n)

    def np_random(self, *a, **k):
        self.calls["np_random"] += 1
        if self.trial:
            return 0.0
        return O.draw_uniform(self.env_q.pop(0))

    # python side (Agent/main.py:35-36)
    def py_random(self):
        self.calls["py_random"] += 1
        return O.draw_uniform(self.agent_x[0])

    def py_randint(self, a, b):
        self.calls["py_randint"] += 1
        assert (a, b) == (0, 3)
        return O.draw_action(self.agent_x[1])


class Injected:
    def __init__(self, env_mod, feed: Feed):
        self.env_mod, self.feed = env_mod, feed

    def __enter__(self):
        f = self.feed
        self.saved = (np.random.randint, np.random.random, pyrandom.random, pyrandom.randint,
                      self.env_mod.Game2048.is_game_over)
        np.random.randint, np.random.random = f.np_randint, f.np_random
        pyrandom.random, pyrandom.randint = f.py_random, f.py_randint
        orig_over = self.saved[4]

        def over(game):
            f.trial = True
            try:
                return orig_over(game)
            finally:
                f.trial = False

        self.env_mod.Game2048.is_game_over = over
        return f

    def __exit__(self, *exc):
        (np.random.randint, np.random.random, pyrandom.random, pyrandom.randint,
         self.env_mod.Game2048.is_game_over) = self.saved


def raw(log2_board):
    return O.to_raw(np.asarray(log2_board).reshape(4, 4))


def lg(raw_board):
    return O.to_log2(np.asarray(raw_board)).reshape(-1)


# --------------------------------------------------------------------------------------
# G1: exhaustive row table through the reference move_left
# --------------------------------------------------------------------------------------
def gen_g1(env_mod):
    feed = Feed()
    n = 16 ** 4
    out_rows = np.zeros((n, 4), dtype=np.uint8)
    out_score = np.zeros(n, dtype=np.int64)
    out_moved = np.zeros(n, dtype=np.uint8)
    with Injected(env_mod, feed):
        feed.env_q = [0] * 8
        g = env_mod.Game2048()
        for idx in range(n):
            v = [(idx >> (4 * c)) & 15 for c in range(4)]
            g.board = np.zeros((4, 4), dtype=int)
            g.board[0] = [0 if x == 0 else 1 << x for x in v]
            moved, score = g.move_left()
            out_rows[idx] = lg(g.board[0])
            out_score[idx] = int(score)
            out_moved[idx] = bool(moved)
            assert not g.board[1:].any()
    np.savez_compressed(os.path.join(HERE, "g1_rows.npz"), rows_out=out_rows,
                        score=out_score, moved=out_moved)
    print("G1 rows:", n, "moved:", int(out_moved.sum()))


def random_boards(rng, count, full_frac=0.2):
    boards = np.zeros((count, 16), dtype=np.uint8)
    for i in range(count):
        kind = rng.random()
        if kind < full_frac:      # full boards over a small alphabet (dead or nearly dead)
            hi = rng.integers(2, 5)
            boards[i] = rng.integers(1, hi + 1, size=16)
        elif kind < full_frac + 0.1:  # checkerboard-like dead boards
            a, b = rng.choice(np.arange(1, 12), size=2, replace=False)
            boards[i] = [(a if (r + c) % 2 == 0 else b) for r in range(4) for c in range(4)]
            if rng.random() < 0.5:  # plant one live pair
                p = rng.integers(0, 15)
                if p % 4 != 3:
                    boards[i][p + 1] = boards[i][p]
        else:
            density = rng.uniform(0.1, 1.0)
            mask = rng.random(16) < density
            vals = np.minimum(rng.geometric(0.35, size=16), 15)
            boards[i] = np.where(mask, vals, 0)
            if not boards[i].any():
                boards[i][rng.integers(0, 16)] = 1
    return boards


# --------------------------------------------------------------------------------------
# G2: Game2048.move (rotate, move_left, rotate back, spawn) on random boards x 4 actions
# --------------------------------------------------------------------------------------
def gen_g2(env_mod, count=10000):
    rng = np.random.default_rng(2048)
    boards = random_boards(rng, count)
    actions = np.tile(np.arange(4, dtype=np.uint8), count // 4 + 1)[:count]
    rng.shuffle(actions)
    dpos = rng.integers(0, 2 ** 32, size=count, dtype=np.uint64).astype(np.uint32)
    dval = rng.integers(0, 2 ** 32, size=count, dtype=np.uint64).astype(np.uint32)
    out = np.zeros_like(boards)
    score = np.zeros(count, dtype=np.int64)
    moved = np.zeros(count, dtype=np.uint8)
    feed = Feed()
    with Injected(env_mod, feed):
        feed.env_q = [0] * 4
        g = env_mod.Game2048()
        for i in range(count):
            g.board = raw(boards[i]).astype(int)
            feed.env_q = [int(dpos[i]), int(dval[i])]
            m, s = g.move(int(actions[i]))
            out[i] = lg(g.board)
            score[i], moved[i] = int(s), bool(m)
    np.savez_compressed(os.path.join(HERE, "g2_moves.npz"), boards=boards, actions=actions,
                        draw_pos=dpos, draw_val=dval, boards_out=out, score=score, moved=moved)
    print("G2 moves:", count, "moved:", int(moved.sum()))


# --------------------------------------------------------------------------------------
# G3: is_game_over
# --------------------------------------------------------------------------------------
def gen_g3(env_mod, count=4000):
    rng = np.random.default_rng(3)
    boards = random_boards(rng, count, full_frac=0.6)
    over = np.zeros(count, dtype=np.uint8)
    feed = Feed()
    with Injected(env_mod, feed):
        feed.env_q = [0] * 4
        g = env_mod.Game2048()
        for i in range(count):
            g.board = raw(boards[i]).astype(int)
            over[i] = bool(g.is_game_over())
            assert np.array_equal(lg(g.board), boards[i])  # probe restores the board
    np.savez_compressed(os.path.join(HERE, "g3_game_over.npz"), boards=boards, over=over)
    print("G3 game_over:", count, "over:", int(over.sum()))


# --------------------------------------------------------------------------------------
# G4: env.step on random env states; calculate_reward table; stall sequence
# --------------------------------------------------------------------------------------
def gen_g4(env_mod, count=6000):
    rng = np.random.default_rng(4)
    boards = random_boards(rng, count, full_frac=0.3)
    # lift some boards into the >=512 branches
    for i in range(0, count, 7):
        boards[i][rng.integers(0, 16)] = rng.integers(8, 13)
    actions = rng.integers(0, 4, size=count).astype(np.uint8)
    dpos = rng.integers(0, 2 ** 32, size=count, dtype=np.uint64).astype(np.uint32)
    dval = rng.integers(0, 2 ** 32, size=count, dtype=np.uint64).astype(np.uint32)
    prev_max = rng.integers(1, 13, size=count).astype(np.int32)          # log2(previous_max)
    cons_action = rng.integers(-1, 4, size=count).astype(np.int32)       # -1 = None
    cons_count = np.where(rng.random(count) < 0.5, rng.integers(0, 12, size=count),
                          rng.integers(8, 130, size=count)).astype(np.int64)
    score0 = rng.integers(0, 5000, size=count).astype(np.int64)
    last_pen = np.zeros(count, dtype=np.float64)
    out = {k: [] for k in ("boards_out", "reward", "done", "max", "score", "prev_max",
                           "cons_action", "cons_count", "last_pen", "valid")}
    feed = Feed()
    with Injected(env_mod, feed):
        for i in range(count):
            feed.env_q = [0] * 4
            e = env_mod.Game2048_env()
            e.game.board = raw(boards[i]).astype(int)
            e.score = int(score0[i])
            e.previous_max = 1 << int(prev_max[i])
            e.consecutive_action = None if cons_action[i] < 0 else int(cons_action[i])
            e.consecutive_count = int(cons_count[i])
            # a penalty state consistent with the streak (reference: -1 * 1.1^k, floor -10)
            p = -1.0
            for _ in range(max(0, int(cons_count[i]) - 10)):
                p = max(p * 1.1, -10)
            e.last_consecutive_penalty = p
            last_pen[i] = p
            before = e.game.board.copy()
            feed.env_q = [int(dpos[i]), int(dval[i])]
            b, r, d, m = e.step(int(actions[i]))
            out["boards_out"].append(lg(b))
            out["reward"].append(float(r))
            out["done"].append(bool(d))
            out["max"].append(int(m))
            out["score"].append(int(e.score))
            out["prev_max"].append(int(np.log2(e.previous_max)))
            out["cons_action"].append(-1 if e.consecutive_action is None
                                      else int(e.consecutive_action))
            out["cons_count"].append(int(e.consecutive_count))
            out["last_pen"].append(float(e.last_consecutive_penalty))
            out["valid"].append(not np.array_equal(before, b))
    np.savez_compressed(
        os.path.join(HERE, "g4_env_step.npz"), boards=boards, actions=actions, draw_pos=dpos,
        draw_val=dval, prev_max_in=prev_max, cons_action_in=cons_action,
        cons_count_in=cons_count, score_in=score0, last_pen_in=last_pen,
        boards_out=np.array(out["boards_out"], dtype=np.uint8),
        reward=np.array(out["reward"], dtype=np.float64),
        done=np.array(out["done"], dtype=np.uint8), max=np.array(out["max"], dtype=np.int64),
        score=np.array(out["score"], dtype=np.int64),
        prev_max=np.array(out["prev_max"], dtype=np.int32),
        cons_action=np.array(out["cons_action"], dtype=np.int32),
        cons_count=np.array(out["cons_count"], dtype=np.int64),
        last_pen=np.array(out["last_pen"], dtype=np.float64),
        valid=np.array(out["valid"], dtype=np.uint8))
    print("G4 env.step:", count, "done:", int(np.sum(out["done"])),
          "valid:", int(np.sum(out["valid"])))

    # calculate_reward table over (score, valid, game_over, L, prev)
    rows = []
    with Injected(env_mod, feed):
        feed.env_q = [0] * 4
        e = env_mod.Game2048_env()
        scores = [0, 4, 8, 12, 36, 100, 256, 1024, 2052, 4096, 16384, 65536, 131072, 262144]
        for L in range(1, 18):
            for prev in range(1, 18):
                for valid in (0, 1):
                    for over in ((0, 1) if not valid else (0,)):
                        for s in (scores if valid else [0]):
                            e.previous_max = 1 << prev
                            r = e.calculate_reward(score=np.int64(s), valid=bool(valid),
                                                   game_over=bool(over),
                                                   max_number=np.int64(1 << L))
                            rows.append((s, valid, over, L, prev, float(r),
                                         int(np.log2(e.previous_max))))
    rows = np.array(rows, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "g4_reward_table.npz"), table=rows)
    print("G4 reward table:", len(rows))

    # stall sequence: an immovable board, action 0 repeated, across a reset
    with Injected(env_mod, feed):
        feed.env_q = [0] * 4
        e = env_mod.Game2048_env()
        stuck = np.array([[2, 0, 0, 0], [4, 0, 0, 0], [2, 0, 0, 0], [4, 0, 0, 0]])
        e.game.board = stuck.copy()
        seq = []
        for t in range(120):
            feed.env_q = [0, 0]
            b, r, d, m = e.step(0)
            seq.append((float(r), bool(d), int(e.consecutive_count),
                        float(e.last_consecutive_penalty)))
        feed.env_q = [0, 0, 0, 0]
        e.reset()
        e.game.board = stuck.copy()
        feed.env_q = [0, 0]
        b, r, d, m = e.step(0)
        after_reset = (float(r), bool(d), int(e.consecutive_count))
        feed.env_q = [0, 0]
        b, r, d, m = e.step(1)  # a different action clears the streak
        after_change = (float(r), bool(d), int(e.consecutive_count),
                        float(e.last_consecutive_penalty))
    with open(os.path.join(HERE, "g4_stall.json"), "w") as fh:
        json.dump({"board": lg(stuck).tolist(), "action": 0, "seq": seq,
                   "after_reset": after_reset, "after_change": after_change}, fh)
    print("G4 stall: first penalised step", next(i for i, s in enumerate(seq) if s[3] < -1) + 1,
          "first done", next(i for i, s in enumerate(seq) if s[1]) + 1)


# --------------------------------------------------------------------------------------
# G5: agent KATs
# --------------------------------------------------------------------------------------
def gen_g5(env_mod, agent_mod):
    A = agent_mod.QLearningAgent
    sched = {}
    for (E, e0, emin) in [(1000, 0.95, 0.01), (3, 1.0, 0.01), (200, 1.0, 0.05), (37, 0.5, 0.02)]:
        a = A(E, 4, exploration_rate=e0, exploration_min=emin)
        eps = []
        for ep in range(E + 5):
            a.decay_exploration(ep)
            eps.append(float(a.epsilon))
        sched[f"{E},{e0},{emin}"] = eps
    rng = np.random.default_rng(5)
    td = []
    for i in range(2000):
        a = A(10, 4, learning_rate=float(rng.choice([0.1, 0.5, 0.01])),
              discount_factor=float(rng.choice([0.9, 0.99, 0.5])))
        s = tuple(map(tuple, raw(random_boards(rng, 1)[0])))
        same = rng.random() < 0.25
        s2 = s if same else tuple(map(tuple, raw(random_boards(rng, 1)[0])))
        qs = rng.normal(0, 3, size=4)
        qs2 = rng.normal(0, 3, size=4)
        if rng.random() < 0.3:
            qs2[rng.integers(0, 4)] = qs2.max()  # force argmax ties
        if rng.random() < 0.1:
            qs2[:] = 0
        a.q_table[s][:] = qs
        if not same:
            a.q_table[s2][:] = qs2
        act = int(rng.integers(0, 4))
        r = float(rng.uniform(-20, 10))
        done = bool(rng.random() < 0.2)
        before_s2 = a.q_table[s2].copy()
        a.update_q_value(s, act, r, s2, done)
        td.append(dict(lr=a.lr, gamma=a.gamma, s=lg(np.array(s)).tolist(),
                       s2=lg(np.array(s2)).tolist(), q_s=qs.tolist(), q_s2=before_s2.tolist(),
                       action=act, reward=r, done=done, q_s_after=a.q_table[s].tolist()))
    # choose_action: explore/exploit and first-max argmax
    ch = []
    feed = Feed()
    with Injected(env_mod, feed):
        for i in range(2000):
            a = A(10, 4, exploration_rate=float(rng.choice([0.0, 0.3, 0.95, 1.0])))
            s = tuple(map(tuple, raw(random_boards(rng, 1)[0])))
            q = np.round(rng.normal(0, 1, size=4), 1)  # rounding makes ties common
            if rng.random() < 0.2:
                q[:] = 0
            a.q_table[s][:] = q
            x0, x1 = (int(v) for v in rng.integers(0, 2 ** 32, size=2, dtype=np.uint64))
            feed.agent_x = (x0, x1)
            act = int(a.choose_action(s))
            ch.append(dict(eps=a.epsilon, s=lg(np.array(s)).tolist(), q=q.tolist(), x0=x0,
                           x1=x1, action=act))
    with open(os.path.join(HERE, "g5_agent.json"), "w") as fh:
        json.dump({"epsilon_schedule": sched, "td": td, "choose": ch}, fh)
    print("G5 agent: schedules", len(sched), "td", len(td), "choose", len(ch))


# --------------------------------------------------------------------------------------
# G6/G7: whole-loop transcripts (Agent/main.py:80-109) with injected counter-RNG draws
# --------------------------------------------------------------------------------------
def run_loop(env_mod, agent_mod, B, seed, env_id0, total_steps=None, episodes=None,
             E=30, eps0=0.95, gamma=0.99, lr=0.1, decay=True):
    """B reference envs, ONE reference agent, lane-sequential inside a step.
    B == 1 with decay=True is exactly the loop of Agent/main.py:80-109 (without the CSV)."""
    feed = Feed()
    rec = {k: [] for k in ("boards", "actions", "rewards", "dones", "maxes")}
    eps_trace, ep_returns, ep_scores = [], [], []
    with Injected(env_mod, feed):
        envs, states, episode_idx, totals = [], [], [0] * B, [0.0] * B
        for i in range(B):
            feed.env_q = [int(v) for v in O.draws(seed, env_id0 + i, 0, O.STREAM_RESET)]
            e = env_mod.Game2048_env()                                   # main.py:66
            envs.append(e)
            states.append(tuple(map(tuple, e.game.board)))               # :81-82
        agent = agent_mod.QLearningAgent(E, 4, learning_rate=lr, discount_factor=gamma,
                                         exploration_rate=eps0)          # :68
        ctr, finished = 0, 0
        while True:
            if total_steps is not None and ctr >= total_steps:
                break
            if episodes is not None and finished >= episodes:
                break
            for i in range(B):
                x = [int(v) for v in O.draws(seed, env_id0 + i, ctr, O.STREAM_STEP)]
                feed.agent_x = (x[0], x[1])
                feed.env_q = [x[2], x[3]]
                e, state = envs[i], states[i]
                rec["boards"].append(lg(np.array(state)))
                action = agent.choose_action(state)                      # :92
                nxt, reward, done, info = e.step(action)                 # :93
                nxt = tuple(map(tuple, nxt))                             # :94
                _ = agent.q_table[state]                                 # :96
                agent.update_q_value(state, action, reward, nxt, done)   # :99
                states[i] = nxt                                          # :100
                totals[i] += reward                                      # :101
                rec["actions"].append(int(action))
                rec["rewards"].append(float(reward))
                rec["dones"].append(bool(done))
                rec["maxes"].append(int(info))
                if done:
                    ep_returns.append(totals[i])
                    ep_scores.append(int(e.score))
                    if decay:
                        agent.decay_exploration(finished)                # :109
                        eps_trace.append(float(agent.epsilon))
                    finished += 1
                    episode_idx[i] += 1
                    feed.env_q = [int(v) for v in
                                  O.draws(seed, env_id0 + i, episode_idx[i], O.STREAM_RESET)]
                    states[i] = tuple(map(tuple, e.reset()))             # :81-82
                    totals[i] = 0.0                                      # :84
            ctr += 1
        keys = np.array([lg(np.array(k)) for k in agent.q_table.keys()], dtype=np.uint8)
        vals = np.array([v for v in agent.q_table.values()], dtype=np.float64)
        final_boards = np.array([lg(e.game.board) for e in envs], dtype=np.uint8)
    return dict(
        B=B, seed=seed, env_id0=env_id0, steps=ctr, E=E, eps0=eps0, gamma=gamma, lr=lr,
        decay=decay, boards=np.array(rec["boards"], dtype=np.uint8),
        actions=np.array(rec["actions"], dtype=np.uint8),
        rewards=np.array(rec["rewards"], dtype=np.float64),
        dones=np.array(rec["dones"], dtype=np.uint8),
        maxes=np.array(rec["maxes"], dtype=np.int64), eps_trace=np.array(eps_trace),
        ep_returns=np.array(ep_returns), ep_scores=np.array(ep_scores, dtype=np.int64),
        q_keys=keys, q_vals=vals, final_boards=final_boards,
        final_epsilon=float(agent.epsilon), calls=json.dumps(feed.calls))


def gen_g6(env_mod, agent_mod):
    for name, kw in {
        "g6_episodes_seed0": dict(B=1, seed=0, env_id0=0, episodes=30, E=30),
        "g6_episodes_seed7": dict(B=1, seed=7, env_id0=123456789, episodes=20, E=20,
                                  eps0=0.5, gamma=0.9),
        "g7_batched_b8": dict(B=8, seed=11, env_id0=1000, total_steps=400, E=1000,
                              eps0=0.3, decay=False),
    }.items():
        tr = run_loop(env_mod, agent_mod, **kw)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **tr)
        print(name, "steps", tr["steps"], "rows", len(tr["q_keys"]),
              "episodes", len(tr["ep_returns"]), "max tile", int(tr["maxes"].max()))


# --------------------------------------------------------------------------------------
# G8: the DQN path's env (Deep_QLearning/environment/Game2048_nopenalty_env.py): step() with
# calculate_reward2 and done = game_over.  It imports gymnasium (stubbed above), numpy, math and
# collections only -- no TensorFlow.  In this env is_game_over's moves are real (their spawn
# draws are consumed and their result stays in moved_board), so nothing is fed as a dummy.
# --------------------------------------------------------------------------------------
def import_reference_dqn_env():
    import_reference()                                   # registers the gymnasium stub
    spec = importlib.util.spec_from_file_location(
        "ref_dqn_env", os.path.join(REF, "Deep_QLearning", "environment", "Game2048_nopenalty_env.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                         # __main__ guard keeps its smoke loop off
    return mod


class InjectedDqn:
    """np.random.randint / np.random.random <- the feed.  The only other patch is a router around
    Game2048.is_game_over: while it runs, draws come from `feed.over_q` instead of `feed.env_q`,
    so that the spawn of the chosen move and the spawn inside is_game_over's move each take the
    pair the build assigns to them whichever of the two happen (behaviour is untouched)."""

    def __init__(self, dqn_mod, feed: Feed):
        self.mod, self.feed = dqn_mod, feed

    def __enter__(self):
        f = self.feed
        f.over_q = []
        self.saved = (np.random.randint, np.random.random, self.mod.Game2048.is_game_over)
        np.random.randint, np.random.random = f.np_randint, f.np_random
        orig_over = self.saved[2]

        def over(game):
            f.env_q, f.over_q = f.over_q, f.env_q
            try:
                return orig_over(game)
            finally:
                f.env_q, f.over_q = f.over_q, f.env_q

        self.mod.Game2048.is_game_over = over
        return f

    def __exit__(self, *exc):
        np.random.randint, np.random.random, self.mod.Game2048.is_game_over = self.saved


def gen_g8(dqn_mod, count=6000, transcript_steps=4000):
    rng = np.random.default_rng(8)
    boards = random_boards(rng, count, full_frac=0.45)   # many full boards: the is_game_over path
    actions = rng.integers(0, 4, size=count).astype(np.uint8)
    draws4 = rng.integers(0, 2 ** 32, size=(count, 4), dtype=np.uint64).astype(np.uint32)
    score0 = rng.integers(0, 5000, size=count).astype(np.int64)
    out = {k: [] for k in ("boards_out", "reward", "done", "max", "score", "used")}
    feed = Feed()
    with InjectedDqn(dqn_mod, feed):
        for i in range(count):
            feed.env_q = [0] * 4
            e = dqn_mod.Game2048_env()
            e.game.board = raw(boards[i]).astype(int)
            e.score = int(score0[i])
            feed.env_q = [int(draws4[i, 0]), int(draws4[i, 1])]      # spawn of the chosen move
            feed.over_q = [int(draws4[i, 2]), int(draws4[i, 3])]     # spawn inside is_game_over
            b, r, d, m = e.step(int(actions[i]))
            out["boards_out"].append(lg(b))
            out["reward"].append(float(r))
            out["done"].append(bool(d))
            out["max"].append(int(m))
            out["score"].append(int(e.score))
            out["used"].append((2 - len(feed.env_q)) + 4 * (2 - len(feed.over_q)))
    # a whole random-play run with the caller's write-back (mainDQL_CNN_step2.py:237
    # `env.game.board = next_state`) and reset on done (:151), driven by the build's counter RNG:
    # action = draw_action(x1); spawn draws x2, x3; is_game_over's spawn: stream 2, words 0, 1
    seed, env_id = 88, 4242
    tb, tr, td, tm, ta = [], [], [], [], []
    with InjectedDqn(dqn_mod, feed):
        feed.env_q = [int(v) for v in O.draws(seed, env_id, 0, O.STREAM_RESET)]
        e = dqn_mod.Game2048_env()
        episode = 0
        board0 = lg(e.game.board)
        for t in range(transcript_steps):
            x = O.draws(seed, env_id, t, O.STREAM_STEP)
            y = O.draws(seed, env_id, t, 2)
            a = O.draw_action(int(x[1]))
            feed.env_q, feed.over_q = [int(x[2]), int(x[3])], [int(y[0]), int(y[1])]
            nb, r, d, m = e.step(a)
            e.game.board = nb                                                   # :237
            ta.append(a); tb.append(lg(nb)); tr.append(float(r)); td.append(bool(d)); tm.append(int(m))
            if d:
                episode += 1
                feed.env_q = [int(v) for v in O.draws(seed, env_id, episode, O.STREAM_RESET)]
                e.reset()                                                       # :151
        final_board = lg(e.game.board)
    np.savez_compressed(
        os.path.join(HERE, "g8_dqn_env.npz"), boards=boards, actions=actions, draws=draws4,
        score_in=score0, boards_out=np.array(out["boards_out"], dtype=np.uint8),
        reward=np.array(out["reward"], dtype=np.float64), done=np.array(out["done"], dtype=np.uint8),
        max=np.array(out["max"], dtype=np.int64), score=np.array(out["score"], dtype=np.int64),
        draws_used=np.array(out["used"], dtype=np.uint8),
        t_seed=seed, t_env_id=env_id, t_board0=board0, t_actions=np.array(ta, dtype=np.uint8),
        t_boards=np.array(tb, dtype=np.uint8), t_reward=np.array(tr, dtype=np.float64),
        t_done=np.array(td, dtype=np.uint8), t_max=np.array(tm, dtype=np.int64),
        t_final_board=final_board, t_episodes=episode)
    print("G8 dqn env:", count, "steps; done", int(np.sum(out["done"])), "draw-pair use {0: none, 2: move, "
          "8: is_game_over, 10: both}", np.bincount(out["used"], minlength=11)[[0, 2, 8, 10]].tolist(),
          "| transcript", transcript_steps, "steps,", episode, "episodes")


def main():
    if "--only-g8" in sys.argv:
        gen_g8(import_reference_dqn_env())
        return
    env_mod, agent_mod = import_reference()
    gen_g8(import_reference_dqn_env())
    gen_g1(env_mod)
    gen_g2(env_mod)
    gen_g3(env_mod)
    gen_g4(env_mod)
    gen_g5(env_mod, agent_mod)
    gen_g6(env_mod, agent_mod)


if __name__ == "__main__":
    main()
