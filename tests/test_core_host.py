"""Checks the per-lane arithmetic the HIP kernels inline (csrc/q2048_core.hpp), compiled for the
host by g++ (tests/hostcheck), against the CPU oracle and the golden vectors -- exhaustively
where the domain is small.  Integer/byte results are bit-exact; rewards use the kernel's own
log2 (relative error < 2^-46) and must round to the same float32 as the reference's float64."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, REPO, load_npz

CSRC = os.path.join(REPO, "2048_q-learning_amd", "csrc")
HC_DIR = os.path.join(REPO, "tests", "hostcheck")


@pytest.fixture(scope="session")
def hc():
    so = os.path.join(HC_DIR, "libhostcheck.so")
    srcs = [os.path.join(HC_DIR, "hostcheck.cpp"), os.path.join(CSRC, "q2048_core.hpp"),
            os.path.join(CSRC, "q2048_luts.inc")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(map(os.path.getmtime, srcs)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                        "-Wall", "-I", CSRC, "-o", so, srcs[0]], check=True)
    L = C.CDLL(so)
    L.hc_mix64.restype = C.c_uint64
    L.hc_mix64.argtypes = [C.c_uint64]
    L.hc_lane_salt.restype = C.c_uint64
    L.hc_lane_salt.argtypes = [C.c_uint64]
    assert L.hc_sizeof_aux() == 16
    return L


def p(a):
    return a.ctypes.data_as(C.c_void_p)


AUX_DTYPE = np.dtype([("score", "<i4"), ("ep_return", "<f4"), ("prev_max", "u1"),
                      ("cons_action", "u1"), ("cons_count", "<u2"), ("episode", "<u4")])


def hc_move(hc, boards, actions):
    boards = np.ascontiguousarray(boards, dtype=np.uint8)
    actions = np.ascontiguousarray(actions, dtype=np.uint8)
    n = len(boards)
    out = np.zeros_like(boards)
    score = np.zeros(n, dtype=np.uint32)
    moved = np.zeros(n, dtype=np.uint8)
    hc.hc_move(p(boards), p(actions), C.c_int64(n), p(out), p(score), p(moved))
    return out, score, moved


def test_luts_are_current(hc):
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import gen_luts

    with open(os.path.join(CSRC, "q2048_luts.inc")) as fh:
        assert fh.read() == gen_luts.render(), "run tools/gen_luts.py > csrc/q2048_luts.inc"
    pw, lg, st = (np.zeros(32) for _ in range(3))
    hc.hc_luts(p(pw), p(lg), p(st))
    want = gen_luts.tables()
    assert pw.tolist() == want[0] and lg.tolist() == want[1] and st.tolist() == want[2]
    assert st[1] == -1.1 and st[25] == -10.0 and st[31] == -10.0


def test_log2_ge1_accuracy(hc):
    """The kernel's log2: relative error < 2^-46 vs libm on [1.03, 2^24] (the reward function
    never passes an argument below 1.05), absolute error < 2^-50 down to 1; table edges too."""
    rng = np.random.default_rng(7)
    xs = np.concatenate([
        2.0 ** rng.uniform(0, 24, size=400000),
        1.0 + rng.uniform(0, 1, size=100000) * 2.0 ** -rng.integers(0, 30, size=100000),
        (1.0 + np.arange(65) / 64.0)[:, None].repeat(3, 1).ravel()
        * np.tile([1 - 2 ** -52, 1.0, 1 + 2 ** -52], 65),          # around every table boundary
        np.arange(1, 70000, dtype=np.float64) + 1.0,                  # score + 1
        1.0 + 0.05 * np.arange(1, 18), 1.0 + 0.1 * np.arange(1, 18),  # the small-reward cases
    ])
    xs = np.ascontiguousarray(xs[xs >= 1.0])
    out = np.zeros_like(xs)
    hc.hc_log2_ge1(p(xs), C.c_int64(len(xs)), p(out))
    want = np.log2(xs)
    assert np.abs(out - want).max() < 2.0 ** -50 * 32                # absolute, results up to 24
    big = xs >= 1.03125
    rel = np.abs(out[big] - want[big]) / np.abs(want[big])
    assert rel.max() < 2.0 ** -46, rel.max()
    assert np.array_equal(out[big].astype(np.float32), want[big].astype(np.float32))


def test_philox_matches_oracle(hc, O):
    rng = np.random.default_rng(0)
    for _ in range(200):
        c = rng.integers(0, 2 ** 32, size=4, dtype=np.uint64).astype(np.uint32)
        k = rng.integers(0, 2 ** 32, size=2, dtype=np.uint64).astype(np.uint32)
        out = np.zeros(4, dtype=np.uint32)
        hc.hc_philox(p(c), p(k), p(out))
        assert out.tolist() == O.philox(c, k).tolist()
    out = np.zeros(4, dtype=np.uint32)
    hc.hc_draws(C.c_uint64((3 << 32) | 5), C.c_uint64((1 << 32) | 7), 9, 1, p(out))
    assert out.tolist() == O.draws((3 << 32) | 5, (1 << 32) | 7, 9, 1).tolist()


def test_move_exhaustive_lines_all_directions(hc):
    """All 16^4 lines (log2 0..15), placed as row 0 / column 0, moved in all 4 directions,
    against the reference's own row table (golden G1)."""
    g = load_npz("g1_rows.npz")
    idx = np.arange(16 ** 4)
    line = np.stack([(idx >> (4 * c)) & 15 for c in range(4)], axis=1).astype(np.uint8)
    res, score, moved = g["rows_out"], g["score"], g["moved"]
    for action in range(4):
        boards = np.zeros((len(idx), 4, 4), dtype=np.uint8)
        want = np.zeros_like(boards)
        if action == 0:      # left: row, cell 0 = col 0
            boards[:, 0, :], want[:, 0, :] = line, res
        elif action == 2:    # right: cell 0 = col 3
            boards[:, 0, ::-1], want[:, 0, ::-1] = line, res
        elif action == 1:    # up: column, cell 0 = row 0
            boards[:, :, 0], want[:, :, 0] = line, res
        else:                # down: cell 0 = row 3
            boards[:, ::-1, 0], want[:, ::-1, 0] = line, res
        out, s, m = hc_move(hc, boards.reshape(-1, 16), np.full(len(idx), action))
        assert np.array_equal(out, want.reshape(-1, 16)), action
        assert np.array_equal(s.astype(np.int64), score), action
        assert np.array_equal(m, moved), action


def test_move_four_lines_independent(hc, O):
    """Random full-range boards (tiles up to 2^17): the 4 lines of a move do not interact."""
    rng = np.random.default_rng(1)
    n = 20000
    boards = np.where(rng.random((n, 16)) < 0.7, rng.integers(1, 18, size=(n, 16)), 0).astype(np.uint8)
    actions = rng.integers(0, 4, size=n).astype(np.uint8)
    out, s, m = hc_move(hc, boards, actions)
    for i in range(n):
        b, sc, mv = O.move(boards[i], int(actions[i]))
        assert out[i].tolist() == b.tolist() and s[i] == sc and bool(m[i]) == mv, i


def test_g2_move_and_spawn(hc):
    g = load_npz("g2_moves.npz")
    out, s, m = hc_move(hc, g["boards"], g["actions"])
    sp = np.zeros_like(out)
    hc.hc_spawn(p(out), p(g["draw_pos"]), p(g["draw_val"]), C.c_int64(len(out)), p(sp))
    final = np.where(m[:, None] != 0, sp, out)
    assert np.array_equal(final, g["boards_out"])
    assert np.array_equal(s.astype(np.int64), g["score"]) and np.array_equal(m, g["moved"])


def test_kth_set_bit_exhaustive(hc):
    masks, ks, want = [], [], []
    for mask in range(1, 1 << 16):
        bits = [i for i in range(16) if mask >> i & 1]
        for k, pos in enumerate(bits):
            masks.append(mask); ks.append(k); want.append(pos)
    masks = np.array(masks, dtype=np.uint16); ks = np.array(ks, dtype=np.uint8)
    out = np.zeros(len(masks), dtype=np.uint8)
    hc.hc_kth_set_bit(p(masks), p(ks), C.c_int64(len(masks)), p(out))
    assert np.array_equal(out, np.array(want, dtype=np.uint8))


def test_spawn_matches_oracle(hc, O):
    rng = np.random.default_rng(2)
    n = 20000
    boards = np.where(rng.random((n, 16)) < rng.random((n, 1)), rng.integers(1, 12, size=(n, 16)), 0).astype(np.uint8)
    boards[:50] = rng.integers(1, 5, size=(50, 16))  # full boards: spawn is a no-op
    xp = rng.integers(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
    xv = rng.integers(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
    xv[:2000] = rng.integers(3865470560, 3865470575, size=2000)  # around the 0.9 threshold
    xp[2000:2100] = 0xFFFFFFFF
    out = np.zeros_like(boards)
    hc.hc_spawn(p(boards), p(xp), p(xv), C.c_int64(n), p(out))
    for i in range(n):
        assert out[i].tolist() == O.add_number(boards[i], int(xp[i]), int(xv[i])).tolist(), i


def _props(hc, boards):
    boards = np.ascontiguousarray(boards, dtype=np.uint8)
    n = len(boards)
    over = np.zeros(n, np.uint8); mx = np.zeros(n, np.uint8); em = np.zeros(n, np.uint16)
    keys = np.zeros(n, np.uint64); ov = np.zeros(n, np.uint8)
    hc.hc_board_props(p(boards), C.c_int64(n), p(over), p(mx), p(em), p(keys), p(ov))
    return over, mx, em, keys, ov


def test_game_over_max_empties_key(hc, O):
    g = load_npz("g3_game_over.npz")
    over, mx, em, keys, ov = _props(hc, g["boards"])
    assert np.array_equal(over, g["over"])                       # the reference's trial-move answer
    assert np.array_equal(mx, g["boards"].max(axis=1))
    want_em = ((g["boards"] == 0) * (1 << np.arange(16))).sum(axis=1)
    assert np.array_equal(em, want_em.astype(np.uint16))
    want_keys = (g["boards"].astype(np.uint64) << (4 * np.arange(16, dtype=np.uint64))).sum(axis=1)
    assert np.array_equal(keys, want_keys) and not ov.any()
    back = np.zeros_like(g["boards"])
    hc.hc_unpack_keys(p(keys), C.c_int64(len(keys)), p(back))
    assert np.array_equal(back, g["boards"])
    # random full boards over a tiny alphabet: closed form == oracle trial moves
    rng = np.random.default_rng(3)
    fb = rng.integers(1, 4, size=(30000, 16)).astype(np.uint8)
    over2 = _props(hc, fb)[0]
    want = np.array([O.is_game_over(b) for b in fb], dtype=np.uint8)
    assert np.array_equal(over2, want) and 0 < want.sum() < len(want)
    big = np.zeros((2, 16), dtype=np.uint8); big[0, 3] = 16; big[1, 7] = 15
    assert _props(hc, big)[4].tolist() == [1, 0]


def _aux_from_golden(g):
    n = len(g["boards"])
    aux = np.zeros(n, dtype=AUX_DTYPE)
    aux["score"] = g["score_in"]
    aux["prev_max"] = g["prev_max_in"]
    aux["cons_action"] = np.where(g["cons_action_in"] < 0, 0xFF, g["cons_action_in"])
    aux["cons_count"] = g["cons_count_in"]
    return aux


def hc_env_step(hc, boards, aux, actions, xp, xv):
    boards = np.ascontiguousarray(boards, dtype=np.uint8).copy()
    aux = aux.copy()
    n = len(boards)
    r32 = np.zeros(n, np.float32); r64 = np.zeros(n, np.float64)
    done = np.zeros(n, np.uint8); mx = np.zeros(n, np.uint8); valid = np.zeros(n, np.uint8)
    score = np.zeros(n, np.uint32)
    hc.hc_env_step(p(boards), p(aux), p(np.ascontiguousarray(actions, dtype=np.uint8)),
                   p(np.ascontiguousarray(xp, dtype=np.uint32)),
                   p(np.ascontiguousarray(xv, dtype=np.uint32)), C.c_int64(n), p(r32), p(r64),
                   p(done), p(mx), p(valid), p(score))
    return boards, aux, r32, r64, done, mx, valid, score


def test_g4_env_step_vs_reference(hc):
    g = load_npz("g4_env_step.npz")
    boards, aux, r32, r64, done, mx, valid, score = hc_env_step(
        hc, g["boards"], _aux_from_golden(g), g["actions"], g["draw_pos"], g["draw_val"])
    assert np.array_equal(boards, g["boards_out"])
    assert np.allclose(r64, g["reward"], rtol=2.0 ** -45, atol=0)  # before the float32 rounding
    assert np.array_equal(r32, g["reward"].astype(np.float32))    # what the TD update consumes
    assert np.array_equal(done, g["done"]) and np.array_equal(valid, g["valid"])
    assert np.array_equal(1 << mx.astype(np.int64), g["max"])
    assert np.array_equal(aux["score"], g["score"])
    assert np.array_equal(aux["prev_max"], g["prev_max"])
    assert np.array_equal(aux["cons_action"], np.where(g["cons_action"] < 0, 0xFF, g["cons_action"]))
    assert np.array_equal(aux["cons_count"], g["cons_count"])


def test_stall_sequence(hc):
    with open(os.path.join(GOLDEN, "g4_stall.json")) as fh:
        st = json.load(fh)
    boards = np.array([st["board"]], dtype=np.uint8)
    aux = np.zeros(1, dtype=AUX_DTYPE)
    aux["prev_max"], aux["cons_action"] = 1, 0xFF
    for t, (r, d, cnt, pen) in enumerate(st["seq"]):
        boards, aux, r32, r64, done, *_ = hc_env_step(hc, boards, aux, [st["action"]], [0], [0])
        assert (r32[0], bool(done[0]), int(aux["cons_count"][0])) == (np.float32(r), d, cnt), t
    # reset keeps the streak: next identical action is done immediately
    aux["score"] = 0
    boards, aux, r32, r64, done, *_ = hc_env_step(hc, boards, aux, [st["action"]], [0], [0])
    assert [r32[0], bool(done[0]), int(aux["cons_count"][0])] == [np.float32(st["after_reset"][0])] + st["after_reset"][1:]
    boards, aux, r32, r64, done, *_ = hc_env_step(hc, boards, aux, [1], [0], [0])
    assert [r32[0], bool(done[0]), int(aux["cons_count"][0])] == [np.float32(st["after_change"][0])] + st["after_change"][1:3]


def test_rollout_env_only_matches_oracle(hc, O):
    """Kernel sequencing (step, auto-reset with per-episode draws) == oracle driver."""
    B, steps, seed, id0, ctr0 = 96, 600, 1234, 5_000_000_000, 77
    rng = np.random.default_rng(4)
    # biased actions so that games end (dead boards + invalid move) and streak rules fire
    actions = np.where(rng.random((steps, B)) < 0.5, rng.integers(0, 2, size=(steps, B)),
                       rng.integers(0, 4, size=(steps, B))).astype(np.uint8)
    actions[:, :8] = 0  # lanes that repeat one action forever: >100-repeat terminations
    envs = O.envs_init(B, 4, seed, id0)
    boards = np.zeros((B, 16), np.uint8); aux = np.zeros(B, dtype=AUX_DTYPE)
    hc.hc_init_envs(p(boards), p(aux), C.c_int64(B), C.c_uint64(seed), C.c_uint64(id0))
    assert np.array_equal(boards, envs["board"][:, :16])
    si, sf, acts, rew, dn = O.rollout(envs, None, steps, seed, id0, ctr0, actions=actions, record=True)
    r32 = np.zeros((steps, B), np.float32); done = np.zeros((steps, B), np.uint8)
    hc.hc_rollout_env(p(boards), p(aux), C.c_int64(B), C.c_int64(steps), C.c_uint64(seed),
                      C.c_uint64(id0), ctr0, p(actions), p(r32), p(done))
    assert np.array_equal(boards, envs["board"][:, :16])
    assert np.array_equal(done, dn) and dn.sum() > 50
    assert np.array_equal(r32, rew.astype(np.float32))
    assert np.array_equal(aux["score"], envs["score"])
    assert np.array_equal(aux["episode"], envs["episode"])
    assert np.array_equal(aux["prev_max"], envs["previous_max_log2"])
    assert np.array_equal(aux["cons_count"], np.minimum(envs["consecutive_count"], 60000))
    assert np.array_equal(aux["cons_action"], envs["consecutive_action"])
    assert np.allclose(aux["ep_return"], envs["episode_return"], rtol=1e-5, atol=1e-5)


def test_eps_greedy_and_td(hc, O):
    with open(os.path.join(GOLDEN, "g5_agent.json")) as fh:
        g5 = json.load(fh)
    ch = g5["choose"]
    n = len(ch)
    eps = np.array([c["eps"] for c in ch]); x0 = np.array([c["x0"] for c in ch], dtype=np.uint32)
    x1 = np.array([c["x1"] for c in ch], dtype=np.uint32)
    q = np.array([c["q"] for c in ch], dtype=np.float32)
    act = np.zeros(n, np.uint8); ex = np.zeros(n, np.uint8)
    hc.hc_eps_greedy(p(eps), p(x0), p(x1), p(q), C.c_int64(n), p(act), p(ex))
    assert act.tolist() == [c["action"] for c in ch]
    td = g5["td"]
    n = len(td)
    q_sa = np.array([c["q_s"][c["action"]] for c in td], dtype=np.float32)
    rw = np.array([c["reward"] for c in td], dtype=np.float32)
    qn = np.array([c["q_s2"] for c in td], dtype=np.float32)
    dn = np.array([c["done"] for c in td], dtype=np.uint8)
    lr = np.array([c["lr"] for c in td]); gm = np.array([c["gamma"] for c in td])
    out = np.zeros(n, np.float32)
    hc.hc_td(p(q_sa), p(rw), p(qn), p(dn), p(lr), p(gm), C.c_int64(n), p(out))
    want = np.array([c["q_s_after"][c["action"]] for c in td])
    # float32 storage of inputs and output: 1e-5 relative (north-star tolerance), 1e-6 absolute
    assert np.allclose(out, want, rtol=1e-5, atol=1e-6)


def test_salt_and_mix(hc):
    seen = {hc.hc_lane_salt(i) for i in range(100000)}
    assert len(seen) == 100000 and all(s & 1 for s in list(seen)[:100])
    assert hc.hc_mix64(0) == 0 and hc.hc_mix64(1) != hc.hc_mix64(2)


# ---------------------------------------------------------------------------------------------
# 5x5 geometry (csrc/q2048_core5.hpp) against the oracle's n-generic restatement (n = 5).  The
# reference hard-codes n = 4; the oracle's generic code is the one pinned to it at n = 4 above.
# ---------------------------------------------------------------------------------------------
def hc5_move(hc, boards, actions):
    boards = np.ascontiguousarray(boards, dtype=np.uint8)
    actions = np.ascontiguousarray(actions, dtype=np.uint8)
    n = len(boards)
    out = np.zeros_like(boards); score = np.zeros(n, np.uint32); moved = np.zeros(n, np.uint8)
    hc.hc5_move(p(boards), p(actions), C.c_int64(n), p(out), p(score), p(moved))
    return out, score, moved


def test_5x5_lines_all_directions(hc, O):
    """Random and structured 5-cell lines (log2 0..26) as row 0 / column 0, 4 directions.
    (The move score is a uint32 on the device: exact while merged tiles stay below 2^30.)"""
    rng = np.random.default_rng(50)
    lines = np.concatenate([
        np.where(rng.random((60000, 5)) < 0.75, rng.integers(1, 5, size=(60000, 5)), 0),  # many merges
        np.where(rng.random((20000, 5)) < 0.6, rng.integers(1, 27, size=(20000, 5)), 0),
        np.array([[1, 1, 1, 1, 1], [2, 2, 2, 2, 0], [0, 0, 0, 0, 3], [1, 1, 2, 2, 3], [26, 26, 0, 26, 26],
                  [1, 0, 1, 0, 1], [5, 5, 5, 0, 5], [0, 0, 0, 0, 0], [1, 2, 3, 4, 5]]),
    ]).astype(np.uint8)
    want_line, want_score, want_moved = [], [], []
    for ln in lines:
        o, s, m = O.move_left_line(ln)
        want_line.append(o); want_score.append(s); want_moved.append(m)
    want_line = np.array(want_line); want_score = np.array(want_score); want_moved = np.array(want_moved)
    n = len(lines)
    for action in range(4):
        boards = np.zeros((n, 5, 5), np.uint8); want = np.zeros_like(boards)
        if action == 0:
            boards[:, 0, :], want[:, 0, :] = lines, want_line
        elif action == 2:
            boards[:, 0, ::-1], want[:, 0, ::-1] = lines, want_line
        elif action == 1:
            boards[:, :, 0], want[:, :, 0] = lines, want_line
        else:
            boards[:, ::-1, 0], want[:, ::-1, 0] = lines, want_line
        out, s, m = hc5_move(hc, boards.reshape(n, 25), np.full(n, action))
        assert np.array_equal(out, want.reshape(n, 25)), action
        assert np.array_equal(s.astype(np.int64), want_score), action
        assert np.array_equal(m.astype(bool), want_moved), action


def test_5x5_moves_spawn_props_match_oracle(hc, O):
    rng = np.random.default_rng(51)
    n = 20000
    boards = np.where(rng.random((n, 25)) < rng.random((n, 1)), rng.integers(1, 8, size=(n, 25)), 0).astype(np.uint8)
    boards[:3000] = rng.integers(1, 13, size=(3000, 25))         # full boards: ~3 % are dead
    boards[3000:3100] = [(1 + (r + c) % 2) for r in range(5) for c in range(5)]   # dead checkerboard
    actions = rng.integers(0, 4, size=n).astype(np.uint8)
    out, s, m = hc5_move(hc, boards, actions)
    xp = rng.integers(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
    xv = rng.integers(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
    sp = np.zeros_like(out)
    hc.hc5_spawn(p(out), p(xp), p(xv), C.c_int64(n), p(sp))
    over = np.zeros(n, np.uint8); mx = np.zeros(n, np.uint8); em = np.zeros(n, np.uint32)
    keys = np.zeros((n, 2), np.uint64); rt = np.zeros_like(boards)
    hc.hc5_board_props(p(boards), C.c_int64(n), p(over), p(mx), p(em), p(keys), p(rt))
    for i in range(n):
        b, sc, mv = O.move(boards[i], int(actions[i]), n=5)
        assert out[i].tolist() == b.tolist() and s[i] == sc and bool(m[i]) == mv, i
        assert sp[i].tolist() == O.add_number(out[i], int(xp[i]), int(xv[i]), n=5).tolist(), i
        if i < 6000:
            assert bool(over[i]) == O.is_game_over(boards[i], n=5), i
    assert over[3000:3100].all() and 0 < over[:3000].sum() < 3000
    assert np.array_equal(mx, boards.max(axis=1))
    assert np.array_equal(em, ((boards == 0) * (1 << np.arange(25))).sum(axis=1).astype(np.uint32))
    assert np.array_equal(rt, boards)                                   # key pack/unpack round trip
    assert len({(int(a), int(b)) for a, b in keys}) == len({bytes(b) for b in boards})
    assert np.all(keys >> np.uint64(63) == 1)


def test_kth_set_bit32(hc):
    rng = np.random.default_rng(52)
    masks = rng.integers(1, 1 << 25, size=200000, dtype=np.uint32)
    masks[:25] = 1 << np.arange(25)
    masks[25] = (1 << 25) - 1
    pc = np.array([bin(int(m)).count("1") for m in masks])
    ks = (rng.random(len(masks)) * pc).astype(np.uint8)
    out = np.zeros(len(masks), np.uint8)
    hc.hc_kth_set_bit32(p(masks), p(ks), C.c_int64(len(masks)), p(out))
    for m, k, o in zip(masks[:5000].tolist(), ks[:5000].tolist(), out[:5000].tolist()):
        assert [i for i in range(25) if m >> i & 1][k] == o
    below = (masks.astype(np.int64) & ((1 << out.astype(np.int64)) - 1))
    assert np.array_equal(np.array([bin(int(x)).count("1") for x in below]), ks)
    assert np.all((masks >> out) & 1)


def test_5x5_rollout_env_only_matches_oracle(hc, O):
    B, steps, seed, id0, ctr0 = 64, 1500, 99, 12345, 5
    rng = np.random.default_rng(53)
    actions = np.where(rng.random((steps, B)) < 0.6, rng.integers(0, 2, size=(steps, B)),
                       rng.integers(0, 4, size=(steps, B))).astype(np.uint8)
    actions[:, :4] = 1
    envs = O.envs_init(B, 5, seed, id0)
    boards = np.zeros((B, 25), np.uint8); aux = np.zeros(B, dtype=AUX_DTYPE)
    hc.hc5_init_envs(p(boards), p(aux), C.c_int64(B), C.c_uint64(seed), C.c_uint64(id0))
    assert np.array_equal(boards, envs["board"][:, :25])
    si, sf, acts, rew, dn = O.rollout(envs, None, steps, seed, id0, ctr0, actions=actions, record=True)
    r32 = np.zeros((steps, B), np.float32); done = np.zeros((steps, B), np.uint8)
    hc.hc5_rollout_env(p(boards), p(aux), C.c_int64(B), C.c_int64(steps), C.c_uint64(seed),
                       C.c_uint64(id0), ctr0, p(actions), p(r32), p(done))
    assert np.array_equal(boards, envs["board"][:, :25])
    assert np.array_equal(done, dn) and dn.sum() > 20
    assert np.array_equal(r32, rew.astype(np.float32))
    assert np.array_equal(aux["score"], envs["score"])
    assert np.array_equal(aux["episode"], envs["episode"])
    assert np.array_equal(aux["prev_max"], envs["previous_max_log2"])


# ---- env profiles: the DQN path's step (Game2048_nopenalty_env.py) and the shaping-state reset ----
def test_g8_dqn_env_step_vs_reference(hc):
    """The device arithmetic of the second env profile against the 6000 reference `step` calls."""
    g = load_npz("g8_dqn_env.npz")
    boards = np.ascontiguousarray(g["boards"], dtype=np.uint8).copy()
    n = len(boards)
    aux = np.zeros(n, dtype=AUX_DTYPE)
    aux["score"], aux["prev_max"], aux["cons_action"] = g["score_in"], 1, 0xFF
    r = np.zeros(n, np.float32); done = np.zeros(n, np.uint8); mx = np.zeros(n, np.uint8)
    valid = np.zeros(n, np.uint8)
    hc.hc_env_step_dqn(p(boards), p(aux), p(np.ascontiguousarray(g["actions"])),
                       p(np.ascontiguousarray(g["draws"], dtype=np.uint32)), C.c_int64(n), 4,
                       p(r), p(done), p(mx), p(valid))
    assert np.array_equal(boards, g["boards_out"])
    assert np.array_equal(r.astype(np.float64), g["reward"])          # scores and -10: exact in f32
    assert np.array_equal(done, g["done"]) and np.array_equal(aux["score"], g["score"])
    assert np.array_equal(np.where(mx > 0, 1 << mx.astype(np.int64), 0), g["max"])
    assert np.array_equal(aux["prev_max"], np.ones(n)) and np.all(aux["cons_count"] == 0)  # untouched


def _hc_rollout_profile(hc, O, side, env, B, steps, seed, id0, actions):
    cells = side * side
    envs = O.envs_init(B, side, seed, id0)
    boards = np.ascontiguousarray(envs["board"][:, :cells]).copy()
    aux = np.zeros(B, dtype=AUX_DTYPE)
    aux["prev_max"], aux["cons_action"] = 1, 0xFF
    rew = np.zeros((steps, B), np.float32); dn = np.zeros((steps, B), np.uint8)
    hc.hc_rollout_env_profile(p(boards), p(aux), C.c_int64(B), side, env, C.c_int64(steps),
                              C.c_uint64(seed), C.c_uint64(id0), C.c_uint32(0), p(actions), p(rew), p(dn))
    si, sf, _, orew, odn = O.rollout(envs, None, steps, seed, id0, 0, actions=actions, record=True,
                                     env_flags=env)
    return boards, aux, rew, dn, envs, orew, odn


@pytest.mark.parametrize("side,env", [(4, 1), (5, 1), (4, 2), (5, 2), (4, 3)])
def test_env_profile_rollouts_match_oracle(hc, O, side, env):
    """Rollouts with resets under every env profile, 4x4 and 5x5: boards, rewards, dones, aux."""
    B, steps, seed, id0 = 96, 700, 11 + env, 5000
    rng = np.random.default_rng(side * 10 + env)
    actions = np.where(rng.random((steps, B)) < 0.5, rng.integers(0, 2, size=(steps, B)),
                       rng.integers(0, 4, size=(steps, B))).astype(np.uint8)
    actions[:, :8] = 3                      # stall lanes: forced terminations (profiles 0 / 2)
    boards, aux, rew, dn, envs, orew, odn = _hc_rollout_profile(hc, O, side, env, B, steps, seed, id0, actions)
    assert np.array_equal(dn, odn) and dn.sum() > 20
    assert np.array_equal(boards, envs["board"][:, :side * side])
    assert np.array_equal(rew, orew.astype(np.float32))
    assert np.array_equal(aux["score"], envs["score"]) and np.array_equal(aux["episode"], envs["episode"])
    if not env & 1:                         # the DQN step keeps no shaping state
        assert np.array_equal(aux["prev_max"], envs["previous_max_log2"])
        assert np.array_equal(aux["cons_count"], np.minimum(envs["consecutive_count"], 60000))
