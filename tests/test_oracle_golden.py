"""Pins the CPU oracle (oracle/q2048_oracle.c) to the golden vectors that
tests/golden/generate_golden.py recorded FROM THE REFERENCE (draw-injected).  Everything here
is bit-exact: boards, scores, flags, float64 rewards and float64 Q-values."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_npz


# ---- RNG known answers (Random123 kat_vectors, philox4x32-10) ---------------------------
def test_philox_known_answers(O):
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
        ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
        ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0],
         [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
    ]
    for ctr, key, want in kat:
        assert [int(v) for v in O.philox(ctr, key)] == want


def test_draw_contract(O):
    assert O.draw_uniform(0) == 0.0
    assert O.draw_uniform(0xFFFFFFFF) == (2 ** 32 - 1) / 2 ** 32 < 1.0
    assert [O.draw_action(x) for x in (0, 0x3FFFFFFF, 0x40000000, 0xFFFFFFFF)] == [0, 0, 1, 3]
    assert O.draw_index(0xFFFFFFFF, 16) == 15 and O.draw_index(0, 16) == 0
    assert O.draw_index(0x80000000, 3) == 1
    # uniform(x) < 0.9  <=>  x < 3865470567
    assert not O.draw_is_four(3865470566) and O.draw_is_four(3865470567)
    d = O.draws(5, 7, 9, O.STREAM_STEP)
    assert [int(v) for v in d] == [int(v) for v in O.philox([7, 0, 9, 0], [5, 0])]
    big = O.draws((3 << 32) | 5, (1 << 32) | 7, 9, O.STREAM_RESET)
    assert [int(v) for v in big] == [int(v) for v in O.philox([7, 1, 9, 1], [5, 3])]


# ---- G1: exhaustive row table --------------------------------------------------------------
def test_g1_all_rows(O):
    g = load_npz("g1_rows.npz")
    for idx in range(16 ** 4):
        row = [(idx >> (4 * c)) & 15 for c in range(4)]
        out, score, moved = O.move_left_line(row)
        assert out.tolist() == g["rows_out"][idx].tolist(), (row, out)
        assert score == g["score"][idx] and moved == bool(g["moved"][idx]), row


def test_row_examples_from_survey(O):
    # SURVEY.md section 9 (Game2048_env.py:28-40): no cascade, one merge per tile
    assert O.move_left_line([1, 1, 1, 0])[0].tolist() == [2, 1, 0, 0]
    assert O.move_left_line([1, 1, 1, 0])[1] == 4
    assert O.move_left_line([2, 2, 3, 3])[0].tolist() == [3, 4, 0, 0]
    assert O.move_left_line([2, 2, 3, 3])[1] == 24
    assert O.move_left_line([1, 1, 2, 3])[0].tolist() == [2, 2, 3, 0]
    assert O.move_left_line([1, 1, 1, 1])[0].tolist() == [2, 2, 0, 0]
    assert O.move_left_line([2, 0, 2, 2])[0].tolist() == [3, 2, 0, 0]
    assert O.move_left_line([1, 2, 3, 4])[2] is False


# ---- G2: 4-direction move + spawn ------------------------------------------------------------
def test_g2_moves(O):
    g = load_npz("g2_moves.npz")
    for i in range(len(g["boards"])):
        b, score, moved = O.move(g["boards"][i], int(g["actions"][i]))
        if moved:
            b = O.add_number(b, int(g["draw_pos"][i]), int(g["draw_val"][i]))
        assert b.tolist() == g["boards_out"][i].tolist(), i
        assert score == g["score"][i] and moved == bool(g["moved"][i]), i


def test_action_map(O):
    # 0 left, 1 up, 2 right, 3 down (Game2048_env.py:54)
    b = np.zeros(16, dtype=np.uint8)
    b[5] = 1  # row 1, col 1
    assert np.flatnonzero(O.move(b, 0)[0]).tolist() == [4]
    assert np.flatnonzero(O.move(b, 1)[0]).tolist() == [1]
    assert np.flatnonzero(O.move(b, 2)[0]).tolist() == [7]
    assert np.flatnonzero(O.move(b, 3)[0]).tolist() == [13]
    with pytest.raises(ValueError):
        O.move(b, 4)


# ---- G3: game over -----------------------------------------------------------------------------
def test_g3_game_over(O):
    g = load_npz("g3_game_over.npz")
    got = np.array([O.is_game_over(b) for b in g["boards"]], dtype=np.uint8)
    assert np.array_equal(got, g["over"])
    assert got.sum() > 100 and (1 - got).sum() > 100


# ---- G4: env.step, reward table, stall sequence -----------------------------------------------
def _env_from(O, g, i):
    e = O.Env(4)
    e.set_board(g["boards"][i])
    r = e.rec
    r["score"][0] = g["score_in"][i]
    r["previous_max_log2"][0] = g["prev_max_in"][i]
    r["consecutive_action"][0] = g["cons_action_in"][i]
    r["consecutive_count"][0] = g["cons_count_in"][i]
    r["last_consecutive_penalty"][0] = g["last_pen_in"][i]
    return e


def test_g4_env_step(O):
    g = load_npz("g4_env_step.npz")
    for i in range(len(g["boards"])):
        e = _env_from(O, g, i)
        b, r, d, m, valid = e.step(int(g["actions"][i]), int(g["draw_pos"][i]),
                                   int(g["draw_val"][i]))
        assert b.tolist() == g["boards_out"][i].tolist(), i
        assert r == g["reward"][i], (i, r, g["reward"][i])          # float64, bit-exact
        assert d == bool(g["done"][i]) and (1 << m) == g["max"][i], i
        assert valid == bool(g["valid"][i]), i
        rec = e.rec
        assert rec["score"][0] == g["score"][i]
        assert rec["previous_max_log2"][0] == g["prev_max"][i]
        assert rec["consecutive_action"][0] == g["cons_action"][i]
        assert rec["consecutive_count"][0] == g["cons_count"][i]
        assert rec["last_consecutive_penalty"][0] == g["last_pen"][i]


def test_g4_reward_table(O):
    t = load_npz("g4_reward_table.npz")["table"]
    e = O.Env(4)
    for s, valid, over, L, prev, want, prev_after in t:
        e.rec["previous_max_log2"][0] = int(prev)
        got = e.calculate_reward(int(s), int(valid), int(over), int(L))
        assert got == want, (s, valid, over, L, prev, got, want)
        assert e.rec["previous_max_log2"][0] == int(prev_after)


def test_g4_reward_kats_from_survey(O):
    # (score, valid, over, max, prev) -> reward, measured on the reference
    kats = [((0, 1, 0, 2, 2), 0.070389327891398), ((4, 1, 0, 4, 2), 2.8673818854419744),
            ((4, 1, 0, 4, 4), 2.350497247084133), ((8, 1, 0, 8, 4), 3.670975448417167),
            ((0, 0, 0, 4, 4), -0.2630344058337938), ((0, 0, 0, 256, 256), -0.8479969065549501),
            ((0, 0, 1, 256, 256), -3.170826331965007), ((0, 0, 1, 512, 512), 3.9036755927305484),
            ((0, 0, 1, 1024, 1024), 4.074585234905427), ((36, 1, 0, 512, 512), 6.030848531038007),
            ((1024, 1, 0, 1024, 512), 10.0)]
    e = O.Env(4)
    for (s, v, o, mx, prev), want in kats:
        e.rec["previous_max_log2"][0] = int(np.log2(prev))
        assert e.calculate_reward(s, v, o, int(np.log2(mx))) == want


def test_g4_stall_sequence(O):
    with open(os.path.join(GOLDEN, "g4_stall.json")) as fh:
        st = json.load(fh)
    e = O.Env(4)
    e.set_board(st["board"])
    for t, (r, d, cnt, pen) in enumerate(st["seq"]):
        _, gr, gd, _, _ = e.step(st["action"], 0, 0)
        assert (gr, gd) == (r, d), t
        assert e.rec["consecutive_count"][0] == cnt
        assert e.rec["last_consecutive_penalty"][0] == pen
    # reset keeps previous_max / consecutive_* (Game2048_env.py:187-191)
    e.reset([0, 0, 0, 0])
    e.set_board(st["board"])
    _, gr, gd, _, _ = e.step(st["action"], 0, 0)
    assert [gr, gd, int(e.rec["consecutive_count"][0])] == st["after_reset"]
    _, gr, gd, _, _ = e.step(1, 0, 0)
    assert [gr, gd, int(e.rec["consecutive_count"][0]),
            float(e.rec["last_consecutive_penalty"][0])] == st["after_change"]


# ---- G5: agent ------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def g5():
    with open(os.path.join(GOLDEN, "g5_agent.json")) as fh:
        return json.load(fh)


def test_g5_epsilon_schedule(O, g5):
    for key, want in g5["epsilon_schedule"].items():
        E, e0, emin = key.split(",")
        a = O.Agent(int(E), 4, exploration_rate=float(e0), exploration_min=float(emin))
        got = []
        for ep in range(len(want)):
            a.decay_exploration(ep)
            got.append(a.epsilon)
        assert got == want, key
    sched = g5["epsilon_schedule"]["1000,0.95,0.01"]
    assert sched[0] == 0.9468833333333333 and sched[299] == 0.015000000000002602
    assert sched[300] == 0.011916666666669268 and sched[800] == 0.01


def test_g5_td_update(O, g5):
    for c in g5["td"]:
        # seed rows through the public surface: lr=1, gamma=0, done -> Q[s][a] = reward
        seed = O.Agent(10, 4, learning_rate=1.0, discount_factor=0.0)
        for k in range(4):
            seed.update_q_value(c["s"], k, c["q_s"][k], c["s"], True)
            if c["s2"] != c["s"]:
                seed.update_q_value(c["s2"], k, c["q_s2"][k], c["s2"], True)
        assert seed.q(c["s"]).tolist() == c["q_s"]
        v = seed._view()  # then switch to the case's hyper-parameters in place
        v.lr, v.gamma = c["lr"], c["gamma"]
        seed.update_q_value(c["s"], c["action"], c["reward"], c["s2"], c["done"])
        assert seed.q(c["s"]).tolist() == c["q_s_after"], c


def test_g5_choose_action(O, g5):
    for c in g5["choose"]:
        a = O.Agent(10, 4, learning_rate=1.0, discount_factor=0.0, exploration_rate=c["eps"])
        for k in range(4):
            a.update_q_value(c["s"], k, c["q"][k], c["s"], True)
        assert a.choose_action(c["s"], c["x0"], c["x1"]) == c["action"], c


def test_td_kat_from_survey(O):
    # SURVEY.md section 8(c) G5
    s = [1, 1] + [0] * 14
    s2 = [2] + [0] * 15
    a = O.Agent(10, 4, learning_rate=1.0, discount_factor=0.0)
    for k, v in enumerate([0.5, 1.5, -0.25, 1.5]):
        a.update_q_value(s2, k, v, s2, True)
    v = a._view()
    v.lr, v.gamma = 0.1, 0.99
    a.update_q_value(s, 2, 2.8673818854419744, s2, False)
    assert a.q(s)[2] == 0.43523818854419744
    a.update_q_value(s, 2, 1.0, s2, True)
    assert a.q(s)[2] == 0.4917143696897777
    a.epsilon = 0.0
    assert a.choose_action(s2, 0, 0) == 1  # argmax tie -> first maximum
    assert O.Agent(10, 4, exploration_rate=0.0).choose_action(s, 0, 0) == 0


# ---- G6 / G7: whole-loop transcripts ------------------------------------------------------------
def _replay(O, tr, decay, storage_f32=False):
    B, seed, id0 = int(tr["B"]), int(tr["seed"]), int(tr["env_id0"])
    envs = O.envs_init(B, 4, seed, id0)
    agent = O.Agent(int(tr["E"]), 4, learning_rate=float(tr["lr"]),
                    discount_factor=float(tr["gamma"]), exploration_rate=float(tr["eps0"]),
                    storage_f32=storage_f32)
    boards, acts, rews, dones, eps = [], [], [], [], []
    finished = 0
    for t in range(int(tr["steps"])):
        boards.append(envs["board"][:, :16].copy())
        si, sf, a, r, d = O.rollout(envs, agent, 1, seed, id0, t, record=True)
        acts.append(a[0]); rews.append(r[0]); dones.append(d[0])
        for i in range(B):
            if d[0, i]:
                if decay:
                    agent.decay_exploration(finished)
                    eps.append(agent.epsilon)
                finished += 1
    return envs, agent, (np.concatenate(boards), np.concatenate(acts), np.concatenate(rews),
                         np.concatenate(dones), np.array(eps))


@pytest.mark.parametrize("name", ["g6_episodes_seed0", "g6_episodes_seed7", "g7_batched_b8"])
def test_g6_transcripts(O, name):
    tr = load_npz(name + ".npz")
    decay = bool(tr["decay"])
    if decay and int(tr["B"]) > 1:
        pytest.skip("decay order is only defined for B=1")
    envs, agent, (boards, acts, rews, dones, eps) = _replay(O, tr, decay)
    assert np.array_equal(boards, tr["boards"])
    assert np.array_equal(acts, tr["actions"])
    assert np.array_equal(rews, tr["rewards"])           # float64 bit-exact
    assert np.array_equal(dones, tr["dones"])
    if decay:
        assert np.array_equal(eps, tr["eps_trace"])
    assert np.array_equal(envs["board"][:, :16], tr["final_boards"])
    assert len(agent) == len(tr["q_keys"])               # same rows as the reference dict
    got = np.stack([agent.q(k) for k in tr["q_keys"]])
    assert np.array_equal(got, tr["q_vals"])             # float64 bit-exact Q-table
    assert dones.sum() == len(tr["ep_returns"])


@pytest.mark.parametrize("name", ["g6_episodes_seed0", "g6_episodes_seed7", "g7_batched_b8"])
def test_f32_storage_option(O, name):
    """The oracle's float32-storage option (orc_agent_t.storage_f32: the device's table type, a
    documented deviation from the reference's float64 dict) against the reference's own transcripts:
    the same actions, boards and dones at every step, float32-rounded rewards, and a final dict
    within the north star's 1e-5 of the reference's -- so it can stand in for the float64 agent where
    argmax ties of a float32 table have to break as they do on the device."""
    tr = load_npz(name + ".npz")
    decay = bool(tr["decay"]) and int(tr["B"]) == 1
    envs, agent, (boards, acts, rews, dones, eps) = _replay(O, tr, decay, storage_f32=True)
    assert np.array_equal(boards, tr["boards"]) and np.array_equal(acts, tr["actions"])
    assert np.array_equal(dones, tr["dones"])
    assert np.array_equal(rews, tr["rewards"].astype(np.float32).astype(np.float64))
    assert len(agent) == len(tr["q_keys"])
    got = np.stack([agent.q(k) for k in tr["q_keys"]])
    assert np.array_equal(got, got.astype(np.float32).astype(np.float64))      # rows hold float32 values
    assert np.allclose(got, tr["q_vals"], rtol=1e-5, atol=1e-6)
    assert not np.array_equal(got, tr["q_vals"])                               # and it is a different table
    # the two-phase driver with B = 1 is the same loop: same table, float32 rounding once per step
    if int(tr["B"]) == 1 and not decay:
        envs2 = O.envs_init(1, 4, int(tr["seed"]), int(tr["env_id0"]))
        a2 = O.Agent(int(tr["E"]), 4, learning_rate=float(tr["lr"]), discount_factor=float(tr["gamma"]),
                     exploration_rate=float(tr["eps0"]), storage_f32=True)
        O.rollout_sync(envs2, a2, int(tr["steps"]), int(tr["seed"]), int(tr["env_id0"]), 0)
        assert np.array_equal(np.stack([a2.q(k) for k in tr["q_keys"]]), got)


def test_random_play_rollout_without_agent(O):
    """orc_rollout with neither agent nor actions = uniformly random play (the device's PLAY_ONLY
    rollout at epsilon = 1): equal to an epsilon = 1 agent's trajectory, single- and multi-threaded."""
    B, steps, seed, id0 = 300, 150, 11, 5
    e1, e2, e3 = (O.envs_init(B, 4, seed, id0) for _ in range(3))
    O.rollout(e1, O.Agent(10, 4, exploration_rate=1.0), steps, seed, id0, 0)
    si, _ = O.rollout(e2, None, steps, seed, id0, 0)
    si3, _ = O.rollout_mt(e3, None, steps, seed, id0, 0, threads=3)
    for k in ("board", "score", "episode", "consecutive_count"):
        assert np.array_equal(e1[k], e2[k]) and np.array_equal(e1[k], e3[k]), k
    assert si[O.ST_STEPS] == si3[O.ST_STEPS] == B * steps and si[O.ST_EPISODES] == si3[O.ST_EPISODES] > 0


# ---- G8: the DQN path's env profile (Game2048_nopenalty_env.py:106-138) -------------------------
def test_g8_dqn_env_steps(O):
    """6000 reference `step` calls (calculate_reward2, done = game_over on the pre-move board, the
    board is_game_over's first legal move leaves behind on a full board): boards, reward, done,
    max tile and score, bit for bit."""
    g = load_npz("g8_dqn_env.npz")
    for i in range(len(g["boards"])):
        e = O.Env(4)
        e.set_board(g["boards"][i])
        e.rec["score"][0] = g["score_in"][i]
        d4 = g["draws"][i]
        b, r, d, m, valid = e.step_dqn(int(g["actions"][i]), int(d4[0]), int(d4[1]), int(d4[2]), int(d4[3]))
        assert b.tolist() == g["boards_out"][i].tolist(), i
        assert r == g["reward"][i] and d == bool(g["done"][i]), i
        assert (1 << m if m else 0) == g["max"][i] and e.rec["score"][0] == g["score"][i], i
    assert g["done"].sum() > 100 and (g["draws_used"] == 10).sum() > 1000   # both draw pairs in play


def test_g8_dqn_env_transcript(O):
    """4000 steps of random play through the reference env with the caller's board write-back
    and resets, against orc_rollout_ex(ORC_ENV_DQN) driven by the same counter draws."""
    g = load_npz("g8_dqn_env.npz")
    seed, id0, steps = int(g["t_seed"]), int(g["t_env_id"]), len(g["t_actions"])
    envs = O.envs_init(1, 4, seed, id0)
    assert envs["board"][0, :16].tolist() == g["t_board0"].tolist()
    for t in range(steps):
        si, sf, a, r, d = O.rollout(envs, None, 1, seed, id0, t, actions=g["t_actions"][t:t + 1],
                                    record=True, env_flags=O.ENV_DQN)
        assert r[0, 0] == g["t_reward"][t] and d[0, 0] == g["t_done"][t], t
        if not d[0, 0]:                       # on done the oracle has already reset in place
            assert envs["board"][0, :16].tolist() == g["t_boards"][t].tolist(), t
    assert envs["board"][0, :16].tolist() == g["t_final_board"].tolist()
    assert int(envs["episode"][0]) == int(g["t_episodes"]) > 10


def test_reset_shaping_option(O):
    """ORC_ENV_RESET_SHAPING: after a forced termination by the stall rule the next episode starts
    with the constructor's shaping state instead of ending on its first repeated action."""
    for flags, expect_done_first in ((0, True), (O.ENV_RESET_SHAPING, False)):
        envs = O.envs_init(1, 4, 3, 0)
        envs["board"][0, :16] = [1, 2, 3, 4] + [0] * 12            # action 0 never moves this row
        acts = np.zeros((102, 1), dtype=np.uint8)
        si, sf, a, r, d = O.rollout(envs, None, 102, 3, 0, 0, actions=acts, record=True,
                                    env_flags=flags)
        assert d[100, 0] == 1 and d[:100].sum() == 0               # done at the 101st repeat
        # step 102 (index 101) is the first step of the next episode, again action 0
        if expect_done_first:
            assert d[101, 0] == 1 and envs["consecutive_count"][0] == 102
        else:
            assert d[101, 0] == 0 and envs["consecutive_count"][0] == 1
            assert envs["previous_max_log2"][0] >= 1
