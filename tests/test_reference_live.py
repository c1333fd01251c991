"""Property tests: the CPU oracle == the reference imported LIVE, on hypothesis-generated boards,
actions and draws.  They run only where /root/reference is mounted (the build container) and
skip everywhere else -- the reference never travels; the committed golden vectors are what pins
the oracle on the GPU box."""
import importlib.util
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from conftest import GOLDEN

REF = os.environ.get("Q2048_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(
    not os.path.exists(os.path.join(REF, "QLearningBase", "environment", "Game2048_env.py")),
    reason="reference not mounted (it never travels to the GPU box)")


@pytest.fixture(scope="module")
def ref():
    spec = importlib.util.spec_from_file_location("generate_golden", os.path.join(GOLDEN, "generate_golden.py"))
    gg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gg)
    env_mod, agent_mod = gg.import_reference()
    return gg, env_mod, agent_mod


tiles = st.integers(min_value=0, max_value=15)
boards = st.lists(tiles, min_size=16, max_size=16).filter(lambda b: any(b))
u32 = st.integers(min_value=0, max_value=2 ** 32 - 1)
CFG = dict(max_examples=300, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])


@settings(**CFG)
@given(board=boards, action=st.integers(0, 3), dpos=u32, dval=u32)
def test_move_and_spawn(ref, O, board, action, dpos, dval):
    gg, env_mod, _ = ref
    feed = gg.Feed()
    with gg.Injected(env_mod, feed):
        feed.env_q = [0] * 4
        g = env_mod.Game2048()
        g.board = gg.raw(board).astype(int)
        feed.env_q = [dpos, dval]
        moved, score = g.move(action)
        want = gg.lg(g.board)
    b, s, m = O.move(board, action)
    if m:
        b = O.add_number(b, dpos, dval)
    assert b.tolist() == want.tolist() and s == int(score) and m == bool(moved)


@settings(**CFG)
@given(board=st.lists(st.integers(1, 4), min_size=16, max_size=16))
def test_game_over_full_boards(ref, O, board):
    gg, env_mod, _ = ref
    feed = gg.Feed()
    with gg.Injected(env_mod, feed):
        feed.env_q = [0] * 4
        g = env_mod.Game2048()
        g.board = gg.raw(board).astype(int)
        want = bool(g.is_game_over())
    assert O.is_game_over(board) == want


@settings(**CFG)
@given(board=boards, action=st.integers(0, 3), dpos=u32, dval=u32, prev=st.integers(1, 12),
       cons_action=st.integers(-1, 3), cons_count=st.integers(0, 130), score0=st.integers(0, 5000))
def test_env_step_reward_done_state(ref, O, board, action, dpos, dval, prev, cons_action, cons_count, score0):
    gg, env_mod, _ = ref
    pen = -1.0
    for _ in range(max(0, cons_count - 10)):
        pen = max(pen * 1.1, -10)
    feed = gg.Feed()
    with gg.Injected(env_mod, feed):
        feed.env_q = [0] * 4
        e = env_mod.Game2048_env()
        e.game.board = gg.raw(board).astype(int)
        e.score, e.previous_max = score0, 1 << prev
        e.consecutive_action = None if cons_action < 0 else cons_action
        e.consecutive_count, e.last_consecutive_penalty = cons_count, pen
        feed.env_q = [dpos, dval]
        b, r, d, m = e.step(action)
        want = (gg.lg(b).tolist(), float(r), bool(d), int(m), int(e.score), int(np.log2(e.previous_max)),
                int(e.consecutive_count), float(e.last_consecutive_penalty))
    o = O.Env(4)
    o.set_board(board)
    rec = o.rec
    rec["score"][0], rec["previous_max_log2"][0] = score0, prev
    rec["consecutive_action"][0], rec["consecutive_count"][0] = cons_action, cons_count
    rec["last_consecutive_penalty"][0] = pen
    gb, gr, gd, gm, _ = o.step(action, dpos, dval)
    got = (gb.tolist(), gr, gd, 1 << gm, int(rec["score"][0]), int(rec["previous_max_log2"][0]),
           int(rec["consecutive_count"][0]), float(rec["last_consecutive_penalty"][0]))
    assert got == want            # float64 reward and penalty bit-exact


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(seed=st.integers(0, 2 ** 32 - 1), env_id=st.integers(0, 2 ** 40), eps=st.sampled_from([0.0, 0.2, 0.9, 1.0]))
def test_short_training_loops(ref, O, seed, env_id, eps):
    """120 steps of Agent/main.py:91-101 with injected draws == the oracle's rollout (dict too)."""
    gg, env_mod, agent_mod = ref
    tr = gg.run_loop(env_mod, agent_mod, B=1, seed=seed, env_id0=env_id, total_steps=120, E=1000,
                     eps0=eps, gamma=0.95, decay=False)
    envs = O.envs_init(1, 4, seed, env_id)
    agent = O.Agent(1000, 4, 0.1, 0.95, eps)
    si, sf, a, r, d = O.rollout(envs, agent, 120, seed, env_id, 0, record=True)
    assert np.array_equal(a[:, 0], tr["actions"]) and np.array_equal(r[:, 0], tr["rewards"])
    assert np.array_equal(d[:, 0], tr["dones"])
    assert np.array_equal(envs["board"][:, :16], tr["final_boards"]) and len(agent) == len(tr["q_keys"])
    assert np.array_equal(np.stack([agent.q(k) for k in tr["q_keys"]]), tr["q_vals"])


# ---- the DQN path's env (Deep_QLearning/environment/Game2048_nopenalty_env.py), live ---------------
@pytest.fixture(scope="module")
def ref_dqn(ref):
    gg = ref[0]
    return gg, gg.import_reference_dqn_env()


# boards weighted towards full ones: is_game_over's own moves only happen there
dqn_boards = st.one_of(boards, st.lists(st.integers(1, 5), min_size=16, max_size=16),
                       st.lists(st.integers(1, 3), min_size=16, max_size=16))


@settings(**CFG)
@given(board=dqn_boards, action=st.integers(0, 3), d0=u32, d1=u32, d2=u32, d3=u32, score0=st.integers(0, 5000))
def test_dqn_env_step(ref_dqn, O, board, action, d0, d1, d2, d3, score0):
    """One step of the reference's second env (calculate_reward2, done = game_over, the board that
    is_game_over leaves in moved_board) == orc_env_step_dqn, both draw pairs routed as the build
    assigns them."""
    gg, dqn_mod = ref_dqn
    feed = gg.Feed()
    with gg.InjectedDqn(dqn_mod, feed):
        feed.env_q = [0] * 4
        e = dqn_mod.Game2048_env()
        e.game.board = gg.raw(board).astype(int)
        e.score = score0
        feed.env_q, feed.over_q = [d0, d1], [d2, d3]
        b, r, d, m = e.step(action)
        want = (gg.lg(b).tolist(), float(r), bool(d), int(m), int(e.score))
    o = O.Env(4)
    o.set_board(board)
    o.rec["score"][0] = score0
    gb, gr, gd, gm, _ = o.step_dqn(action, d0, d1, d2, d3)
    assert (gb.tolist(), gr, gd, (1 << gm) if gm else 0, int(o.rec["score"][0])) == want
