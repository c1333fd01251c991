"""GPU parity tests (-m gpu): the HIP path, called through the C ABI (ctypes -> libq2048_hip.so),
against the CPU oracle and the committed golden vectors.

Bars
  * boards, actions, done flags, scores, counters: bit-exact.
  * rewards: the device computes the reference's float64 expression with its own log2 and
    rounds to float32; required |device - float32(oracle)| <= 1 float32 ulp, and the test
    reports how many are not bit-equal (expected: 0 or a handful).
  * Q-values (float32 table vs the reference's float64 dict): rtol 1e-5 (north-star tolerance)
    + atol 1e-6.
Nothing here reads /root/reference."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_npz

pytestmark = pytest.mark.gpu

DEV = "cuda:0"      # tests/test_host_twin.py loads this file a second time with DEV = "cpu" (the CPU twin,
                    # libq2048_host.so) and runs the tests it lists there -- in the CPU suite, without a GPU


def LIB(pkg):
    """The native library of the device under test."""
    return pkg._native.lib_for(torch.device(DEV))


def sync():
    if DEV != "cpu":
        torch.cuda.synchronize()


def release_cached_device_memory():
    """Tests that start child processes on the same GPU first hand this process's cached blocks
    (earlier tests leave up to 128 GiB of freed tables in torch's allocator) back to the device."""
    import gc

    gc.collect()
    if DEV != "cpu":
        torch.cuda.empty_cache()


def ulp32(x):
    return np.spacing(np.abs(np.asarray(x, dtype=np.float32)).astype(np.float32))


def assert_rewards(dev_r32, oracle_r64, what=""):
    want = np.asarray(oracle_r64, dtype=np.float64).astype(np.float32)
    got = np.asarray(dev_r32, dtype=np.float32)
    bad = np.abs(got.astype(np.float64) - want.astype(np.float64)) > ulp32(want)
    assert not bad.any(), f"{what}: {bad.sum()} rewards off by more than 1 ulp"
    n_ne = int((got != want).sum())
    if n_ne:
        print(f"[reward] {what}: {n_ne}/{got.size} differ from float32(oracle) by 1 ulp")
    return n_ne


def oracle_aux(envs):
    return dict(score=envs["score"], prev_max=envs["previous_max_log2"],
                cons_action=envs["consecutive_action"] & 0xFF,
                cons_count=np.minimum(envs["consecutive_count"], 60000), episode=envs["episode"])


def assert_aux(aux_fields, envs, what=""):
    want = oracle_aux(envs)
    for k, v in want.items():
        assert np.array_equal(aux_fields[k].astype(np.int64), np.asarray(v, dtype=np.int64)), (what, k)
    assert np.allclose(aux_fields["ep_return"], envs["episode_return"], rtol=1e-5, atol=1e-4), what


def t8(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint8)).to(DEV)


# ---------------------------------------------------------------------------------------------
# env
# ---------------------------------------------------------------------------------------------
def test_native_library_is_loaded(pkg):
    lib = LIB(pkg)
    assert lib.q2048_abi_version() == pkg._native.ABI_VERSION
    assert os.path.samefile(pkg._native.LIB_PATH, os.path.join(os.path.dirname(pkg.__file__),
                                                                "csrc", "libq2048_hip.so"))


@pytest.mark.parametrize("B,seed,id0", [(1, 0, 0), (777, 3, 10), (65536, 99, (1 << 33) + 5)])
def test_env_init_matches_oracle(pkg, O, B, seed, id0):
    env = pkg.BatchedGame2048Env(B, seed=seed, env_id0=id0, device=DEV)
    envs = O.envs_init(B, 4, seed, id0)
    assert np.array_equal(env.boards.cpu().numpy(), envs["board"][:, :16])
    assert_aux(env.aux_fields(), envs, "init")


def _step_draws(pkg, boards, aux, actions, dpos, dval):
    n = len(boards)
    tb, ta = t8(boards), torch.from_numpy(aux.view(np.uint8).reshape(n, 16).copy()).to(DEV)
    act = t8(actions)
    dp32 = torch.from_numpy(np.ascontiguousarray(dpos, dtype=np.uint32).view(np.int32)).to(DEV)
    dv32 = torch.from_numpy(np.ascontiguousarray(dval, dtype=np.uint32).view(np.int32)).to(DEV)
    r = torch.empty(n, dtype=torch.float32, device=DEV)
    d = torch.empty(n, dtype=torch.uint8, device=DEV)
    m = torch.empty(n, dtype=torch.uint8, device=DEV)
    st = torch.zeros(1, dtype=torch.int32, device=DEV)
    N = pkg._native
    N.check(LIB(pkg).q2048_env_step_draws(tb.data_ptr(), ta.data_ptr(), act.data_ptr(),
                                         dp32.data_ptr(), dv32.data_ptr(), n, 4, r.data_ptr(),
                                         d.data_ptr(), m.data_ptr(), st.data_ptr(), None),
            "env_step_draws")
    sync()
    return (tb.cpu().numpy(), ta.cpu().numpy().view(pkg.AUX_DTYPE).reshape(-1), r.cpu().numpy(),
            d.cpu().numpy(), m.cpu().numpy(), int(st.item()))


def test_golden_g2_moves_on_device(pkg):
    """10k reference moves (all 4 directions) with the spawn draws injected."""
    g = load_npz("g2_moves.npz")
    n = len(g["boards"])
    aux = np.zeros(n, dtype=pkg.AUX_DTYPE)
    aux["prev_max"], aux["cons_action"] = 1, 0xFF
    b, a, r, d, m, st = _step_draws(pkg, g["boards"], aux, g["actions"], g["draw_pos"], g["draw_val"])
    assert st == 0
    assert np.array_equal(b, g["boards_out"])
    assert np.array_equal(a["score"], g["score"])           # env.score after one move = merge score
    assert np.array_equal(m, g["boards_out"].max(axis=1))


def test_golden_g4_env_step_on_device(pkg):
    """6k reference env.step() calls from arbitrary env states: board, reward, done, max, state."""
    g = load_npz("g4_env_step.npz")
    n = len(g["boards"])
    aux = np.zeros(n, dtype=pkg.AUX_DTYPE)
    aux["score"] = g["score_in"]
    aux["prev_max"] = g["prev_max_in"]
    aux["cons_action"] = np.where(g["cons_action_in"] < 0, 0xFF, g["cons_action_in"])
    aux["cons_count"] = g["cons_count_in"]
    b, a, r, d, m, st = _step_draws(pkg, g["boards"], aux, g["actions"], g["draw_pos"], g["draw_val"])
    assert st == 0
    assert np.array_equal(b, g["boards_out"])
    assert np.array_equal(d, g["done"])
    assert np.array_equal(1 << m.astype(np.int64), g["max"])
    assert_rewards(r, g["reward"], "G4")
    assert np.array_equal(a["score"], g["score"])
    assert np.array_equal(a["prev_max"], g["prev_max"])
    assert np.array_equal(a["cons_action"], np.where(g["cons_action"] < 0, 0xFF, g["cons_action"]))
    assert np.array_equal(a["cons_count"], g["cons_count"])


def test_exhaustive_lines_on_device(pkg):
    """All 16^4 lines x 4 directions against the reference's row table (golden G1)."""
    g = load_npz("g1_rows.npz")
    idx = np.arange(16 ** 4)
    line = np.stack([(idx >> (4 * c)) & 15 for c in range(4)], axis=1).astype(np.uint8)
    for action in range(4):
        boards = np.zeros((len(idx), 4, 4), dtype=np.uint8)
        want = np.zeros_like(boards)
        if action == 0:
            boards[:, 0, :], want[:, 0, :] = line, g["rows_out"]
        elif action == 2:
            boards[:, 0, ::-1], want[:, 0, ::-1] = line, g["rows_out"]
        elif action == 1:
            boards[:, :, 0], want[:, :, 0] = line, g["rows_out"]
        else:
            boards[:, ::-1, 0], want[:, ::-1, 0] = line, g["rows_out"]
        aux = np.zeros(len(idx), dtype=pkg.AUX_DTYPE)
        aux["prev_max"], aux["cons_action"] = 17, 0xFF
        # a full-board-safe spawn is irrelevant here: compare only unmoved cells + score
        b, a, r, d, m, st = _step_draws(pkg, boards.reshape(-1, 16), aux,
                                        np.full(len(idx), action), np.zeros(len(idx), np.uint32),
                                        np.zeros(len(idx), np.uint32))
        moved = g["moved"].astype(bool)
        assert np.array_equal(a["score"], g["score"]), action
        wantf = want.reshape(-1, 16)
        assert np.array_equal(b[~moved], wantf[~moved]), action
        # moved boards: exactly one spawned tile (log2 1: draw 0 -> "2") on a formerly empty cell
        diff = b[moved] != wantf[moved]
        assert np.all(diff.sum(axis=1) == 1), action
        assert np.all(b[moved][diff] == 1) and np.all(wantf[moved][diff] == 0), action


def test_stall_sequence_on_device(pkg):
    with open(os.path.join(GOLDEN, "g4_stall.json")) as fh:
        st = json.load(fh)
    boards = np.array([st["board"]], dtype=np.uint8)
    aux = np.zeros(1, dtype=pkg.AUX_DTYPE)
    aux["prev_max"], aux["cons_action"] = 1, 0xFF
    rewards, dones = [], []
    z = np.zeros(1, np.uint32)
    for t in range(len(st["seq"])):
        boards, aux, r, d, m, _ = _step_draws(pkg, boards, aux, [st["action"]], z, z)
        rewards.append(r[0]); dones.append(bool(d[0]))
    assert dones == [s[1] for s in st["seq"]]
    assert_rewards(np.array(rewards), np.array([s[0] for s in st["seq"]]), "stall")
    aux["score"] = 0  # reset() keeps the streak: the next identical action ends the episode
    boards, aux, r, d, m, _ = _step_draws(pkg, boards, aux, [st["action"]], z, z)
    assert bool(d[0]) == st["after_reset"][1] and int(aux["cons_count"][0]) == st["after_reset"][2]


def test_bad_action_is_rejected_not_masked(pkg):
    env = pkg.BatchedGame2048Env(64, seed=1, device=DEV)
    before = env.boards.clone()
    acts = torch.zeros(64, dtype=torch.uint8, device=DEV)
    acts[5] = 4
    acts[9] = 255
    env.step(acts)
    with pytest.raises(ValueError):
        env.check_status()
    after = env.boards.cpu().numpy()
    assert np.array_equal(after[5], before.cpu().numpy()[5])
    assert np.array_equal(after[9], before.cpu().numpy()[9])
    one = pkg.Game2048_env(device=DEV)
    with pytest.raises(ValueError):
        one.step(7)


def test_abi_argument_errors(pkg):
    N = pkg._native
    L = LIB(pkg)
    b = torch.zeros((4, 16), dtype=torch.uint8, device=DEV)
    a = torch.zeros((4, 16), dtype=torch.uint8, device=DEV)
    assert L.q2048_env_init(None, a.data_ptr(), 4, 4, 0, 0, None) == -1
    assert L.q2048_env_init(b.data_ptr(), a.data_ptr(), -1, 4, 0, 0, None) == -2
    assert L.q2048_env_init(b.data_ptr() + 1, a.data_ptr(), 4, 4, 0, 0, None) == -3
    assert L.q2048_env_init(b.data_ptr(), a.data_ptr(), 4, 6, 0, 0, None) == -4
    assert L.q2048_env_init(b.data_ptr(), a.data_ptr(), 0, 4, 0, 0, None) == 0   # empty batch
    assert b"NULL" in L.q2048_strerror(-1)
    with pytest.raises(NotImplementedError):
        pkg.BatchedGame2048Env(4, board_size=6, device=DEV)
    with pytest.raises(RuntimeError):                    # "cuda[:i]" or, by explicit request, "cpu": nothing else
        pkg.BatchedGame2048Env(4, device="meta")
    # a batch whose grid would not fit HIP's 32-bit grid.x is a size error, not a truncated launch
    too_big = (2 ** 31 - 1) * 256 + 1
    assert L.q2048_env_init(b.data_ptr(), a.data_ptr(), too_big, 4, 0, 0, None) == -2
    assert L.q2048_legal_moves(b.data_ptr(), too_big, 4, a.data_ptr(), None) == -2
    assert L.q2048_encode_onehot(b.data_ptr(), (2 ** 31 - 1) * 4 + 1, 0, a.data_ptr(), None) == -2
    with pytest.raises(ValueError):
        pkg.BatchedGame2048Env(4, profile="other", device=DEV)
    # q2048_rowcache_rebind: arguments first; a record that is no visit row of `from_table` is emptied, not re-bound
    c = torch.full((4, 32), 0x5A, dtype=torch.uint8, device=DEV)
    t = torch.zeros(1 << 9, dtype=torch.uint8, device=DEV)            # (stands for two tables of 2^4 slots)
    assert L.q2048_rowcache_rebind(None, 4, 4, t.data_ptr(), 4, t.data_ptr() + 256, 4, None) == -1
    assert L.q2048_rowcache_rebind(c.data_ptr(), 4, 4, None, 4, t.data_ptr(), 4, None) == -1
    assert L.q2048_rowcache_rebind(c.data_ptr(), 4, 4, t.data_ptr(), 3, t.data_ptr(), 4, None) == -2
    assert L.q2048_rowcache_rebind(c.data_ptr() + 8, 4, 4, t.data_ptr(), 4, t.data_ptr() + 256, 4, None) == -3
    assert L.q2048_rowcache_rebind(c.data_ptr(), 4, 6, t.data_ptr(), 4, t.data_ptr() + 256, 4, None) == -4
    assert int(c.min()) == 0x5A
    assert L.q2048_rowcache_rebind(c.data_ptr(), 4, 4, t.data_ptr(), 4, t.data_ptr() + 256, 4, None) == 0
    assert int(c.max()) == 0


def test_flag_bits_outside_the_abi_are_refused(pkg):
    """The shipped library takes the Q2048_FLAG_* bits of include/q2048.h and nothing else (the
    ablation / sort-width / write-mode bits exist only in the measurement build): any other bit is
    Q2048_ERR_FLAGS, nothing is launched.  The deterministic step has no evaluation or learner-less
    form and refuses NO_LEARN / PLAY_ONLY instead of learning anyway."""
    N = pkg._native
    env = pkg.BatchedGame2048Env(64, seed=1, device=DEV)
    agent = pkg.BatchedQLearningAgent(10, capacity_log2=12, seed=1, device=DEV)
    before = env.boards.clone()
    for bits in (1 << 8, 1 << 12, 1 << 14, 5 << 16, 1 << 25, 1 << 31):   # (bit 24 is Q2048_FLAG_LINE_SUMMARY since ABI 7)
        agent.experiment_bits = bits
        with pytest.raises(N.NativeError, match="flag bits"):
            agent.fused_rollout(env, 3)
        with pytest.raises(N.NativeError, match="flag bits"):
            agent.deterministic_rollout(env, 1)
    agent.experiment_bits = 0
    for refused in (N.FLAG_NO_LEARN, N.FLAG_PLAY_ONLY):
        agent.flags = refused
        with pytest.raises(N.NativeError, match="flag bits"):
            agent.deterministic_rollout(env, 1)
    agent.flags = 0
    a = torch.zeros(64, dtype=torch.uint8, device=DEV)
    assert LIB(pkg).q2048_q_choose(agent.table.data_ptr(), 12, env.boards.data_ptr(), 64, 4, 0.5, 1, 0, 0, 1 << 9,
                                  a.data_ptr(), agent.status.data_ptr(), None) == -7
    assert torch.equal(env.boards, before) and agent.table_size() == 0 and env.ctr == 0 == agent.ctr
    # ... and the measurement build of the same sources takes them
    X = N.load(N.build_experiments())
    assert X.q2048_q_choose(agent.table.data_ptr(), 12, env.boards.data_ptr(), 64, 4, 0.5, 1, 0, 0, 1 << 9,
                            a.data_ptr(), agent.status.data_ptr(), None) == 0
    sync()


# ---------------------------------------------------------------------------------------------
# env profiles: the DQN path's env (Game2048_nopenalty_env.py) and the shaping-state reset
# ---------------------------------------------------------------------------------------------
def test_golden_g8_dqn_env_on_device(pkg):
    """6000 reference `step` calls of Deep_QLearning/environment/Game2048_nopenalty_env.py
    (calculate_reward2, done = game_over, the is_game_over quirks) with both draw pairs injected,
    through q2048_env_step_ex(Q2048_FLAG_ENV_DQN)."""
    g = load_npz("g8_dqn_env.npz")
    n = len(g["boards"])
    N = pkg._native
    aux = np.zeros(n, dtype=pkg.AUX_DTYPE)
    aux["score"], aux["prev_max"], aux["cons_action"] = g["score_in"], 1, 0xFF
    tb, ta = t8(g["boards"]), torch.from_numpy(aux.view(np.uint8).reshape(n, 16).copy()).to(DEV)
    d4 = torch.from_numpy(np.ascontiguousarray(g["draws"], dtype=np.uint32).view(np.int32)).to(DEV)
    r = torch.empty(n, dtype=torch.float32, device=DEV)
    d = torch.empty(n, dtype=torch.uint8, device=DEV)
    m = torch.empty(n, dtype=torch.uint8, device=DEV)
    st = torch.zeros(1, dtype=torch.int32, device=DEV)
    N.check(LIB(pkg).q2048_env_step_ex(tb.data_ptr(), ta.data_ptr(), t8(g["actions"]).data_ptr(), n, 4,
                                      0, 0, 0, N.FLAG_ENV_DQN, d4.data_ptr(), r.data_ptr(),
                                      d.data_ptr(), m.data_ptr(), st.data_ptr(), None), "env_step_ex")
    sync()
    assert int(st.item()) == 0
    assert np.array_equal(tb.cpu().numpy(), g["boards_out"])
    assert np.array_equal(r.cpu().numpy().astype(np.float64), g["reward"])   # scores and -10: exact
    assert np.array_equal(d.cpu().numpy(), g["done"])
    mm = m.cpu().numpy().astype(np.int64)
    assert np.array_equal(np.where(mm > 0, 1 << mm, 0), g["max"])
    a = ta.cpu().numpy().view(pkg.AUX_DTYPE).reshape(-1)
    assert np.array_equal(a["score"], g["score"])
    assert np.all(a["prev_max"] == 1) and np.all(a["cons_count"] == 0)       # no shaping state


@pytest.mark.parametrize("n,profile,reset_shaping", [(4, "nopenalty", False), (5, "nopenalty", False),
                                                      (4, "shaped", True), (5, "shaped", True),
                                                      (4, "nopenalty", True)])
def test_env_profiles_fused_and_unfused_match_oracle(pkg, O, n, profile, reset_shaping):
    """Every env profile through BOTH device paths -- the fused rollout and the 4-call API -- on
    independent lanes (eps = 0.25: argmax over learnt rows) against per-lane oracle agents:
    boards and aux bit-exact, every Q row within rtol 1e-5.  Lanes stuck on one action make the
    stall rule end episodes, which is where the shaping-state reset differs from the reference."""
    B, steps, seed, id0, eps, lr, gamma = 160, 500, 41, 77000, 0.25, 0.1, 0.95
    flags = (O.ENV_DQN if profile == "nopenalty" else 0) | (O.ENV_RESET_SHAPING if reset_shaping else 0)
    envs = O.envs_init(B, n, seed, id0)
    agents = [O.Agent(100, 4, lr, gamma, eps, n=n) for _ in range(B)]
    episodes = 0
    for i in range(B):
        si, sf = O.rollout(envs[i:i + 1], agents[i], steps, seed, id0 + i, 0, env_flags=flags)
        episodes += int(si[O.ST_EPISODES])
    assert episodes > 50

    def mk():
        e = pkg.BatchedGame2048Env(B, board_size=n, seed=seed, env_id0=id0, device=DEV,
                                   profile=profile, reset_shaping_state=reset_shaping)
        a = pkg.BatchedQLearningAgent(100, learning_rate=lr, discount_factor=gamma,
                                      exploration_rate=eps, capacity_log2=19, seed=seed, env_id0=id0,
                                      device=DEV, independent=True, board_size=n)
        return e, a

    e1, a1 = mk()
    for k in (1, 199, 300):
        a1.fused_rollout(e1, k)
    e2, a2 = mk()
    _unfused_loop(pkg, e2, a2, steps)
    for what, e, a in (("fused", e1, a1), ("unfused", e2, a2)):
        assert np.array_equal(e.boards.cpu().numpy(), envs["board"][:, :n * n]), what
        f = e.aux_fields()
        assert np.array_equal(f["score"], envs["score"]) and np.array_equal(f["episode"], envs["episode"]), what
        if profile == "shaped":
            assert_aux(f, envs, what)
        for i, oa in enumerate(agents):
            keys, vals = oa.dump()
            got = a.q_values(t8(keys), env_id=id0 + i).cpu().numpy()
            assert np.allclose(got, vals, rtol=1e-5, atol=1e-6 * max(1.0, float(np.abs(vals).max()))), (what, i)
        assert a.table_size() == sum(len(oa) for oa in agents), what
        assert a.check_status() == 0 and e.check_status() == 0
    assert a1.stats()["episodes"] == episodes


def test_play_only_never_touches_the_table(pkg, O):
    """Q2048_FLAG_PLAY_ONLY at eps = 1 = uniformly random play: the same boards as the learner at
    eps = 1 (actions come from the draws alone), a table that stays all zeros, no drops."""
    B, steps, seed, id0 = 5000, 130, 6, 99
    env = pkg.BatchedGame2048Env(B, seed=seed, env_id0=id0, device=DEV)
    agent = pkg.BatchedQLearningAgent(100, exploration_rate=1.0, capacity_log2=16, seed=seed,
                                      env_id0=id0, device=DEV)          # far too small to learn in
    agent.fused_rollout(env, 30, play_only=True)
    agent.fused_rollout(env, steps - 30, play_only=True)
    envs = O.envs_init(B, 4, seed, id0)
    si, sf = O.rollout(envs, O.Agent(100, 4, 0.1, 0.9, 1.0), steps, seed, id0, 0)
    assert np.array_equal(env.boards.cpu().numpy(), envs["board"][:, :16])
    assert_aux(env.aux_fields(), envs, "play only")
    assert int(torch.count_nonzero(agent.table)) == 0 and agent.table_size() == 0
    st = agent.stats()
    assert st["steps"] == B * steps and st["episodes"] == si[O.ST_EPISODES]
    assert st["inserts"] == 0 and st["drops"] == 0 and agent.check_status() == 0
    assert np.isclose(st["reward_sum"], sf[O.SF_REWARD], rtol=1e-5)


def _env_rollout_device(pkg, B, steps, seed, id0, actions):
    env = pkg.BatchedGame2048Env(B, seed=seed, env_id0=id0, device=DEV)
    rew, dn = [], []
    acts = torch.from_numpy(actions).to(DEV)
    for t in range(steps):
        _, r, d, _ = env.step(acts[t])
        rew.append(r.clone()); dn.append(d.clone())
        env.reset(d)
    return env, torch.stack(rew).cpu().numpy(), torch.stack(dn).cpu().numpy()


def test_env_rollout_matches_oracle(pkg, O):
    """step / reset(done) over many steps with given actions (ragged batch: B not a multiple of
    64 or 256), including dead boards, stall terminations and resets."""
    B, steps, seed, id0 = 1000, 400, 42, 7_000_000_000
    rng = np.random.default_rng(6)
    actions = np.where(rng.random((steps, B)) < 0.5, rng.integers(0, 2, size=(steps, B)),
                       rng.integers(0, 4, size=(steps, B))).astype(np.uint8)
    actions[:, :16] = 3
    envs = O.envs_init(B, 4, seed, id0)
    si, sf, _, rew, dn = O.rollout(envs, None, steps, seed, id0, 0, actions=actions, record=True)
    env, drew, ddn = _env_rollout_device(pkg, B, steps, seed, id0, actions)
    assert np.array_equal(ddn.astype(np.uint8), dn) and dn.sum() > 100
    assert np.array_equal(env.boards.cpu().numpy(), envs["board"][:, :16])
    assert_rewards(drew, rew, "env rollout")
    assert_aux(env.aux_fields(), envs, "env rollout")


# ---------------------------------------------------------------------------------------------
# agent
# ---------------------------------------------------------------------------------------------
def test_golden_g5_choose_on_device(pkg):
    with open(os.path.join(GOLDEN, "g5_agent.json")) as fh:
        ch = json.load(fh)["choose"]
    N = pkg._native
    # group by epsilon (a launch takes one epsilon)
    for eps in sorted({c["eps"] for c in ch}):
        cs = [c for c in ch if c["eps"] == eps]
        n = len(cs)
        agent = pkg.BatchedQLearningAgent(10, capacity_log2=14, device=DEV, learning_rate=1.0,
                                          discount_factor=0.0)
        boards = t8(np.array([c["s"] for c in cs]))
        # Q[s][a] = value via lr = 1, gamma = 0, done (exact float32 of the golden value)
        for k in range(4):
            agent.update_q_value(boards, np.full(n, k), np.array([c["q"][k] for c in cs], np.float32),
                                 boards, np.ones(n, bool))
        x0 = torch.from_numpy(np.array([c["x0"] for c in cs], np.uint32).view(np.int32)).to(DEV)
        x1 = torch.from_numpy(np.array([c["x1"] for c in cs], np.uint32).view(np.int32)).to(DEV)
        acts = torch.empty(n, dtype=torch.uint8, device=DEV)
        N.check(LIB(pkg).q2048_q_choose_draws(agent.table.data_ptr(), agent.capacity_log2,
                                             boards.data_ptr(), x0.data_ptr(), x1.data_ptr(), n, 4,
                                             float(eps), 0, 0, acts.data_ptr(),
                                             agent.status.data_ptr(), None), "q_choose_draws")
        # duplicate states inside the golden list share a row; the last write wins in both worlds
        # only if states are unique, so compare on unique states
        keys = [tuple(c["s"]) for c in cs]
        uniq = [i for i, k in enumerate(keys) if keys.count(k) == 1]
        got = acts.cpu().numpy()
        assert [int(got[i]) for i in uniq] == [cs[i]["action"] for i in uniq]


def test_golden_g5_td_on_device(pkg):
    with open(os.path.join(GOLDEN, "g5_agent.json")) as fh:
        td = json.load(fh)["td"]
    for (lr, gamma) in sorted({(c["lr"], c["gamma"]) for c in td}):
        cs = [c for c in td if (c["lr"], c["gamma"]) == (lr, gamma)]
        # keep cases whose states do not collide with another case's states
        seen = {}
        for c in cs:
            for s in (tuple(c["s"]), tuple(c["s2"])):
                seen[s] = seen.get(s, 0) + 1
        cs = [c for c in cs if seen[tuple(c["s"])] == (2 if c["s"] == c["s2"] else 1)
              and seen[tuple(c["s2"])] == (2 if c["s"] == c["s2"] else 1)]
        n = len(cs)
        assert n > 50
        agent = pkg.BatchedQLearningAgent(10, capacity_log2=15, device=DEV, learning_rate=1.0,
                                          discount_factor=0.0)
        s = t8(np.array([c["s"] for c in cs])); s2 = t8(np.array([c["s2"] for c in cs]))
        ones = np.ones(n, bool)
        for k in range(4):
            agent.update_q_value(s2, np.full(n, k), np.array([c["q_s2"][k] for c in cs], np.float32), s2, ones)
            agent.update_q_value(s, np.full(n, k), np.array([c["q_s"][k] for c in cs], np.float32), s, ones)
        agent.lr, agent.gamma = lr, gamma
        agent.update_q_value(s, np.array([c["action"] for c in cs]),
                             np.array([c["reward"] for c in cs], np.float32), s2,
                             np.array([c["done"] for c in cs]))
        got = agent.q_values(s).cpu().numpy()
        want = np.array([c["q_s_after"] for c in cs])
        assert np.allclose(got, want, rtol=1e-5, atol=1e-6)
        assert agent.check_status() == 0


def _unfused_loop(pkg, env, agent, steps, record=False):
    """The reference loop body (Agent/main.py:91-101) through the 4-call batched API."""
    acts, rews, dones = [], [], []
    for _ in range(steps):
        state = env.boards.clone()
        a = agent.choose_action(state)
        nxt, r, d, _ = env.step(a)
        agent.update_q_value(state, a, r, nxt, d)
        if record:
            acts.append(a.clone()); rews.append(r.clone()); dones.append(d.clone())
        env.reset(d)
    if record:
        return (torch.stack(acts).cpu().numpy(), torch.stack(rews).cpu().numpy(),
                torch.stack(dones).cpu().numpy())


@pytest.mark.parametrize("eps,gamma", [(0.3, 0.99), (0.05, 0.9)])
def test_single_env_loop_matches_oracle(pkg, O, eps, gamma):
    """B = 1 == the reference agent exactly (sequential semantics): same actions, same boards,
    Q-table within float32 tolerance -- through the unfused 4-call API."""
    steps, seed, id0 = 1500, 5, 31337
    env = pkg.BatchedGame2048Env(1, seed=seed, env_id0=id0, device=DEV)
    agent = pkg.BatchedQLearningAgent(100, learning_rate=0.1, discount_factor=gamma,
                                      exploration_rate=eps, capacity_log2=16, seed=seed,
                                      env_id0=id0, device=DEV)
    acts, rews, dones = _unfused_loop(pkg, env, agent, steps, record=True)
    envs = O.envs_init(1, 4, seed, id0)
    oa = O.Agent(100, 4, 0.1, gamma, eps)
    si, sf, a, r, d = O.rollout(envs, oa, steps, seed, id0, 0, record=True)
    assert np.array_equal(acts, a) and np.array_equal(dones.astype(np.uint8), d)
    assert np.array_equal(env.boards.cpu().numpy(), envs["board"][:, :16])
    assert_rewards(rews, r, "single env")
    keys, vals = oa.dump()
    got = agent.q_values(t8(keys)).cpu().numpy()
    assert np.allclose(got, vals, rtol=1e-5, atol=1e-6)
    # same rows as the reference's defaultdict: every state update_q_value touched (s and s')
    assert agent.table_size() == len(oa)


def _oracle_independent(O, B, steps, seed, id0, eps, lr, gamma):
    """B independent reference agents, one per env (what FLAG_INDEPENDENT means)."""
    envs = O.envs_init(B, 4, seed, id0)
    agents = [O.Agent(100, 4, lr, gamma, eps) for _ in range(B)]
    tot_i = np.zeros(O.ST_NI, np.int64); tot_f = np.zeros(O.SF_NF)
    for i in range(B):
        si, sf = O.rollout(envs[i:i + 1], agents[i], steps, seed, id0 + i, 0)
        tot_i += si; tot_f += sf
    return envs, agents, tot_i, tot_f


def _check_independent(pkg, O, env, agent, envs, agents, what):
    assert np.array_equal(env.boards.cpu().numpy(), envs["board"][:, :16]), what
    assert_aux(env.aux_fields(), envs, what)
    worst = 0.0
    for i, oa in enumerate(agents):
        keys, vals = oa.dump()
        got = agent.q_values(t8(keys), env_id=agent.env_id0 + i).cpu().numpy()
        assert np.allclose(got, vals, rtol=1e-5, atol=1e-6), (what, i)
        worst = max(worst, float(np.max(np.abs(got - vals) / (np.abs(vals) + 1e-1))))
    print(f"[q] {what}: worst relative Q error {worst:.2e}")


def test_fused_rollout_matches_oracle_independent_lanes(pkg, O):
    """The fused kernel on 200 lanes with private rows == 200 reference agents, bit-exact boards,
    Q within tolerance; statistics equal."""
    B, steps, seed, id0, eps, lr, gamma = 200, 300, 17, 123456, 0.2, 0.1, 0.99
    envs, agents, oi, of = _oracle_independent(O, B, steps, seed, id0, eps, lr, gamma)
    env = pkg.BatchedGame2048Env(B, seed=seed, env_id0=id0, device=DEV)
    agent = pkg.BatchedQLearningAgent(100, learning_rate=lr, discount_factor=gamma,
                                      exploration_rate=eps, capacity_log2=18, seed=seed,
                                      env_id0=id0, device=DEV, independent=True)
    agent.fused_rollout(env, steps)
    _check_independent(pkg, O, env, agent, envs, agents, "fused")
    st = agent.stats()
    assert st["steps"] == B * steps == oi[O.ST_STEPS]
    assert st["episodes"] == oi[O.ST_EPISODES] and st["episodes"] > 50
    assert st["valid_moves"] == oi[O.ST_VALID] and st["explored"] == oi[O.ST_EXPLORE]
    assert st["score_sum"] == oi[O.ST_SCORE] and st["drops"] == 0 and st["cas_retries"] == 0
    hist = {1 << k: int(v) for k, v in enumerate(oi[O.ST_HIST0:O.ST_HIST0 + 24]) if v}
    assert st["max_tile_hist"] == hist
    assert np.isclose(st["return_sum"], of[O.SF_RETURN], rtol=1e-5)
    assert np.isclose(st["reward_sum"], of[O.SF_REWARD], rtol=1e-5)
    assert st["inserts"] == agent.table_size() == sum(len(oa) for oa in agents)
    assert agent.check_status() == 0


@pytest.mark.parametrize("steps_per_launch", [1, 7, 50])
def test_strict_td_mode_equals_default_on_private_rows(pkg, steps_per_launch):
    """Q2048_FLAG_TD_CAS (compare-and-swap TD write) and the default store write give bit-equal
    tables when lanes own their rows, for any split of the rollout into launches."""
    B, steps, seed, id0 = 257, 100, 3, 77
    tabs = []
    for strict in (False, True):
        e = pkg.BatchedGame2048Env(B, seed=seed, env_id0=id0, device=DEV)
        a = pkg.BatchedQLearningAgent(100, exploration_rate=0.2, discount_factor=0.99,
                                      capacity_log2=17, seed=seed, env_id0=id0, device=DEV,
                                      independent=True, strict_td=strict)
        left = steps
        while left > 0:
            k = min(steps_per_launch if strict else steps, left)
            a.fused_rollout(e, k)
            left -= k
        k_, q_ = a.export_rows()
        o = np.argsort(k_)
        tabs.append((e.boards.cpu().numpy(), k_[o], q_[o], a.stats()))
    assert np.array_equal(tabs[0][0], tabs[1][0])
    assert np.array_equal(tabs[0][1], tabs[1][1]) and np.array_equal(tabs[0][2], tabs[1][2])
    for k in ("steps", "episodes", "valid_moves", "score_sum", "inserts", "explored", "drops"):
        assert tabs[0][3][k] == tabs[1][3][k], k
    assert tabs[1][3]["cas_retries"] == 0


def test_fused_equals_unfused_and_split_launches(pkg, O):
    """One K-step launch == K one-step launches == the 4-call API (independent lanes)."""
    B, steps, seed, id0, eps = 300, 120, 8, 999, 0.25

    def mk():
        e = pkg.BatchedGame2048Env(B, seed=seed, env_id0=id0, device=DEV)
        a = pkg.BatchedQLearningAgent(100, exploration_rate=eps, discount_factor=0.95,
                                      capacity_log2=18, seed=seed, env_id0=id0, device=DEV,
                                      independent=True)
        return e, a

    e1, a1 = mk(); a1.fused_rollout(e1, steps)
    e2, a2 = mk()
    for _ in range(steps):
        a2.fused_rollout(e2, 1)
    e3, a3 = mk(); _unfused_loop(pkg, e3, a3, steps)
    assert torch.equal(e1.boards, e2.boards) and torch.equal(e1.boards, e3.boards)
    assert torch.equal(e1.aux, e2.aux)
    f1, f3 = e1.aux_fields(), e3.aux_fields()
    for k in ("score", "prev_max", "cons_action", "cons_count", "episode"):
        assert np.array_equal(f1[k], f3[k]), k
    k1, q1 = a1.export_rows(); k2, q2 = a2.export_rows(); k3, q3 = a3.export_rows()
    o1, o2, o3 = np.argsort(k1), np.argsort(k2), np.argsort(k3)
    assert np.array_equal(k1[o1], k2[o2]) and np.array_equal(k1[o1], k3[o3])
    assert np.array_equal(q1[o1], q2[o2]) and np.array_equal(q1[o1], q3[o3])   # bit-equal tables
    s1, s2 = a1.stats(), a2.stats()
    for k in ("steps", "episodes", "valid_moves", "score_sum", "inserts", "explored"):
        assert s1[k] == s2[k], k


def test_step_outputs_alias_by_default_and_copy_on_request(pkg):
    """`env.step` returns VIEWS of the env's output buffers (no torch kernel in the batched loop):
    the next step overwrites reward / done / max_tile, and the boards returned by step t are
    overwritten by step t + 2 (two ping-pong buffers).  That is the documented contract; a caller that
    keeps outputs across steps asks for copies (`copy_outputs=True`), and then what it kept from step t
    is still step t's after step t + 1."""
    B = 4096
    acts = [torch.full((B,), a, dtype=torch.uint8, device=DEV) for a in (0, 1, 2, 3)]
    e = pkg.BatchedGame2048Env(B, seed=3, device=DEV)
    b1, r1, d1, m1 = e.step(acts[0])
    keep = (b1.clone(), r1.clone(), d1.clone(), m1.clone())
    b2, r2, d2, m2 = e.step(acts[1])
    assert (r2.data_ptr(), d2.data_ptr(), m2.data_ptr()) == (r1.data_ptr(), d1.data_ptr(), m1.data_ptr())
    assert b2.data_ptr() != b1.data_ptr() and torch.equal(b1, keep[0])          # step t's boards survive ONE step
    keep2 = (b2.clone(), r2.clone(), d2.clone(), m2.clone())
    b3, _, _, _ = e.step(acts[2])
    assert b3.data_ptr() == b1.data_ptr()
    c = pkg.BatchedGame2048Env(B, seed=3, device=DEV, copy_outputs=True)
    cb1, cr1, cd1, cm1 = c.step(acts[0])
    cb2, cr2, cd2, cm2 = c.step(acts[1])
    assert torch.equal(cr1, keep[1]) and torch.equal(cd1, keep[2]) and torch.equal(cm1, keep[3])
    assert cr2.data_ptr() != cr1.data_ptr() and cd2.data_ptr() != cd1.data_ptr() and cm2.data_ptr() != cm1.data_ptr()
    assert torch.equal(cb1, keep[0]) and torch.equal(cb2, keep2[0])
    assert torch.equal(cr2, keep2[1]) and torch.equal(cd2, keep2[2]) and torch.equal(cm2, keep2[3])


@pytest.mark.parametrize("n", [4, 5])
def test_fused_rollout_row_cache_and_statistics_mirror(pkg, O, n):
    """q2048_fused_rollout_opts: (a) K steps cut into launches of 1, 7 and 20 steps WITH the row cache
    (a launch starts from the record its predecessor left, not from a probe) equal the single K-step
    launch and the cut launches WITHOUT the cache bit for bit -- boards, aux, tables, statistics;
    (b) the cache is the 4-call API's: fused launches and choose / step / update / reset iterations
    interleaved on one agent equal the same interleaving without a cache; (c) the statistics mirror a
    launch's last block writes to pinned host memory equals the device vectors after every launch, and a
    reader that did not wait for the right launch is refused; (d) argument errors of the opts struct."""
    B, steps, seed, id0, eps = 777, 140, 21, 4242, 0.25      # 777: a ragged last block

    def mk(cache):
        e = pkg.BatchedGame2048Env(B, board_size=n, seed=seed, env_id0=id0, device=DEV)
        a = pkg.BatchedQLearningAgent(100, exploration_rate=eps, discount_factor=0.95, capacity_log2=18,
                                      seed=seed, env_id0=id0, device=DEV, independent=True, board_size=n,
                                      row_cache=cache)
        return e, a

    def cut(e, a, pattern):
        left, k = steps, 0
        while left > 0:
            s_ = min(pattern[k % len(pattern)], left)
            a.fused_rollout(e, s_)
            sync()
            mi, mf = a.mirrored_stats()                                             # (c)
            assert np.array_equal(mi, a.stats_i.cpu().numpy()) and np.array_equal(mf, a.stats_f.cpu().numpy())
            left -= s_; k += 1

    def table(a):
        k, q = a.export_rows()
        o = np.lexsort(k.reshape(len(q), -1).T[::-1])
        return k[o], q[o]

    e0, a0 = mk(True); a0.fused_rollout(e0, steps)
    ref_k, ref_q = table(a0)
    for cache in (True, False):
        for pattern in ((1,), (7,), (20, 3, 1)):
            e, a = mk(cache); cut(e, a, pattern)
            assert torch.equal(e.boards, e0.boards) and torch.equal(e.aux, e0.aux), (cache, pattern)
            k, q = table(a)
            assert np.array_equal(k, ref_k) and np.array_equal(q, ref_q), (cache, pattern)
            s0, s1 = a0.stats(), a.stats()
            for key in ("steps", "episodes", "valid_moves", "score_sum", "inserts", "explored", "drops"):
                assert s0[key] == s1[key], (cache, pattern, key)
            if cache:      # a record holds the key of the board the env is on, unless its row does not exist yet
                rec = a._row_cache.cpu().numpy()
                assert (rec.view(np.uint64)[:, 0] != 0).mean() > 0.95
    # (b) fused launches and 4-call iterations interleaved, with and without the shared cache
    outs = []
    for cache in (True, False):
        e, a = mk(cache)
        s = e.boards
        for rnd in range(6):
            a.fused_rollout(e, 9)
            s = e.boards
            for _ in range(5):
                act = a.choose_action(s)
                s2, r, d, _info = e.step(act)
                a.update_q_value(s, act, r, s2, d)
                s = e.reset(d)
        outs.append((e.boards.clone(), e.aux.clone(), table(a), a.stats()["inserts"], a.table_size()))
        assert a.check_status() == 0
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][2][0], outs[1][2][0]) and np.array_equal(outs[0][2][1], outs[1][2][1])
    assert outs[0][3] == outs[1][3] == outs[0][4]
    # (c) a stale mirror is refused
    e, a = mk(True); a.fused_rollout(e, 3); sync(); a.mirrored_stats()
    a._mirror_launches += 1
    with pytest.raises(RuntimeError):
        a.mirrored_stats()
    # (d) the opts struct: another layout, a mirror without its ticket or without both vectors, alignment
    L, Nn = LIB(pkg), pkg._native
    e, a = mk(True)

    def call(opts, stats_f=True):
        return L.q2048_fused_rollout_opts(
            e.boards.data_ptr(), e.aux.data_ptr(), a.table.data_ptr(), a.capacity_log2, B, n, 1, 0.5, 0.1, 0.9,
            seed, id0, 0, Nn.FLAG_INDEPENDENT, a.stats_i.data_ptr(), a.stats_f.data_ptr() if stats_f else None,
            a.status.data_ptr(), C.byref(opts) if opts is not None else None, None)

    bad = Nn.RolloutOpts(); bad.size = 8
    assert call(bad) == -2
    assert call(Nn.RolloutOpts(stats_mirror=a._mirror.data_ptr())) == -1
    assert call(Nn.RolloutOpts(stats_mirror=a._mirror.data_ptr(), mirror_ticket=a._mirror_ticket.data_ptr()),
                stats_f=False) == -1
    assert call(Nn.RolloutOpts(row_cache=a._cache(B).data_ptr() + 8)) == -3
    assert call(Nn.RolloutOpts(stats_mirror=a._mirror.data_ptr() + 4, mirror_ticket=a._mirror_ticket.data_ptr())) == -3
    assert call(None) == 0 and call(Nn.RolloutOpts()) == 0
    sync()


@pytest.mark.parametrize("n", [4, 5])
def test_row_cache_records_are_bound_to_their_table(pkg, n):
    """A row-cache record carries a tag of the table it was read from: a caller that moves to another
    table -- a growth through the C API, a second table -- and hands the old cache over gets misses, not writes
    through stale slot indices.  The table is swapped behind the agent's back (what a C caller that forgot to
    zero the cache does): the next launch must probe, create its rows in the NEW table and leave no Q value in a
    slot that has no key."""
    B = 4096
    env = pkg.BatchedGame2048Env(B, board_size=n, seed=21, device=DEV)
    agent = pkg.BatchedQLearningAgent(100, exploration_rate=0.5, capacity_log2=18, seed=21, device=DEV, board_size=n)
    agent.fused_rollout(env, 8)
    old = agent.table                                       # stays alive: the new table has another address
    agent.table = torch.zeros_like(old)
    assert agent.table.data_ptr() != old.data_ptr()
    agent.stats(reset=True)
    agent._inserts_folded = 0
    agent._rebase_rows(0)
    agent.fused_rollout(env, 8)                             # the cache still holds the old table's records
    words = agent.table.view(torch.int64).reshape(-1, 4)
    orphan = (words[:, 0] == 0) & ((words[:, 1] != 0) | (words[:, 2] != 0))
    assert int(orphan.sum()) == 0                           # no Q value without a key
    st = agent.stats()
    assert agent.table_size() == st["inserts"] > B and st["drops"] == 0
    _, found = agent.q_values(env.boards, return_found=True)
    carried = agent._row_cache.view(torch.int64).reshape(B, -1)[:, 0] != 0      # envs whose row exists already
    assert bool(found[carried].all())
    del old


@pytest.mark.parametrize("n", [4, 5])
def test_four_call_loop_without_copies_and_row_cache(pkg, O, n):
    """The batched loop of Agent/main.py:92-100 written WITHOUT a board copy (step ping-pongs two
    buffers: the tensor that was env.boards stays the pre-step state) and with the row cache
    (q2048_q_update_cached / q2048_q_choose_cached: s of step t + 1 is s' of step t) equals, bit for
    bit, the same loop with clones and no cache, and the fused rollout; eps = 0.25 so that greedy
    lanes read cached rows, 150 steps so that resets invalidate records.  Then the cache under
    misuse: states passed in another order than the env's lanes -- a record is used only on a key
    match, so the result is still the uncached one."""
    B, steps, seed, id0, eps = 300, 150, 8, 999, 0.25

    def mk(cache):
        e = pkg.BatchedGame2048Env(B, board_size=n, seed=seed, env_id0=id0, device=DEV)
        a = pkg.BatchedQLearningAgent(100, exploration_rate=eps, discount_factor=0.95, capacity_log2=18,
                                      seed=seed, env_id0=id0, device=DEV, independent=True, board_size=n,
                                      row_cache=cache)
        return e, a

    e1, a1 = mk(False); _unfused_loop(pkg, e1, a1, steps)          # clones, no cache
    e2, a2 = mk(True)
    s = e2.boards                      # what the constructor dealt (a reset() here would start episode 1)
    for _ in range(steps):
        a = a2.choose_action(s)
        before = s.clone()
        s2, r, d, info = e2.step(a)
        assert s2.data_ptr() != s.data_ptr() and torch.equal(s, before)      # the state is intact
        assert torch.equal(info, torch.where(e2.max_log2 > 0, 1 << e2.max_log2.int(), 0).int())
        a2.update_q_value(s, a, r, s2, d)
        s = e2.reset(d)
    e3, a3 = mk(True); a3.fused_rollout(e3, steps)
    assert torch.equal(e1.boards, e2.boards) and torch.equal(e1.aux, e2.aux) and torch.equal(e1.boards, e3.boards)
    rows = [a.export_rows() for a in (a1, a2, a3)]
    srt = [np.lexsort(k.reshape(len(q), -1).T[::-1]) for k, q in rows]
    for (k, q), o in zip(rows[1:], srt[1:]):
        assert np.array_equal(k[o], rows[0][0][srt[0]]) and np.array_equal(q[o], rows[0][1][srt[0]])
    s1, s2_ = a1.stats(), a2.stats()
    assert s1["inserts"] == s2_["inserts"] == a2.table_size() and a2.check_status() == 0
    # misuse: every update is fed a random permutation of the lanes' transitions (private rows per
    # lane POSITION, so the uncached result is well defined): records rarely match, and when they do
    # they hold exactly that position's row
    def shuffled(cache):
        e = pkg.BatchedGame2048Env(B, board_size=n, seed=3, device=DEV)
        a = pkg.BatchedQLearningAgent(100, exploration_rate=0.5, capacity_log2=18, seed=3, device=DEV,
                                      board_size=n, row_cache=cache, independent=True)
        g = torch.Generator(device="cpu"); g.manual_seed(5)
        for t in range(60):
            st = e.boards.clone()
            act = a.choose_action(st)
            nx, r, d, _ = e.step(act)
            perm = (torch.randperm(B, generator=g) if t % 3 else torch.arange(B)).to(DEV)
            a.update_q_value(st[perm], act[perm], r[perm], nx[perm], d[perm])
            e.reset(d)
        return e.boards.clone(), a.export_rows()
    (b1, (k1, q1)), (b2, (k2, q2)) = shuffled(False), shuffled(True)
    o1 = np.lexsort(k1.reshape(len(q1), -1).T[::-1]); o2 = np.lexsort(k2.reshape(len(q2), -1).T[::-1])
    assert torch.equal(b1, b2) and np.array_equal(k1[o1], k2[o2]) and np.array_equal(q1[o1], q2[o2])


def test_env_step_to_two_buffers(pkg):
    """q2048_env_step_to: in place == two buffers; the input buffer is untouched; a rejected action
    copies its board; overlapping buffers are refused; max_tile is the raw tile of max_log2."""
    N, L = pkg._native, LIB(pkg)
    B, seed = 1000, 9
    env = pkg.BatchedGame2048Env(B, seed=seed, device=DEV)
    agent = pkg.BatchedQLearningAgent(10, exploration_rate=1.0, capacity_log2=4, seed=seed, device=DEV)
    agent.fused_rollout(env, 60, play_only=True)
    b0, aux0 = env.boards.clone(), env.aux.clone()
    acts = torch.randint(0, 4, (B,), dtype=torch.uint8, device=DEV)
    acts[7] = 9                                                     # rejected
    outs = []
    for two in (False, True):
        bi, aux = b0.clone(), aux0.clone()
        bo = torch.full_like(bi, 255) if two else bi
        r = torch.zeros(B, device=DEV); d = torch.zeros(B, dtype=torch.uint8, device=DEV)
        m = torch.zeros(B, dtype=torch.uint8, device=DEV); mt = torch.zeros(B, dtype=torch.int32, device=DEV)
        st = torch.zeros(1, dtype=torch.int32, device=DEV)
        assert L.q2048_env_step_to(bi.data_ptr(), bo.data_ptr(), aux.data_ptr(), acts.data_ptr(), B, 4, seed, 0, 60,
                                   0, r.data_ptr(), d.data_ptr(), m.data_ptr(), mt.data_ptr(), st.data_ptr(), None) == 0
        sync()
        if two:
            assert torch.equal(bi, b0)
        assert int(st.item()) == N.STATUS_BAD_ACTION and torch.equal(bo[7], b0[7])
        outs.append((bo.clone(), aux, r, d, m, mt))
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    m, mt = outs[0][4], outs[0][5]
    assert torch.equal(mt, torch.where(m > 0, 1 << m.int(), 0).int()) and int(mt.max()) >= 8
    big = torch.zeros((B + 8, 16), dtype=torch.uint8, device=DEV)
    assert L.q2048_env_step_to(big.data_ptr(), big.data_ptr() + 64, aux0.data_ptr(), acts.data_ptr(), B, 4, seed, 0,
                               0, 0, outs[0][2].data_ptr(), outs[0][3].data_ptr(), m.data_ptr(), None,
                               env.status.data_ptr(), None) == -3


@pytest.mark.parametrize("n,load", [(4, 0.5), (4, 0.8), (4, 0.93), (5, 0.75), (5, 0.93)])
def test_bucketised_probing_at_high_load(pkg, n, load):
    """The probe sequence stays inside the 128-byte line of four slots before it moves to the next
    line.  Random keys imported up to load 0.93 of a small table (clusters hundreds of slots long: the
    probe limit is 2^14 slots or the whole table, not round 3's 256 -- at which a racing import at load
    0.93 dropped rows and the case was taken out of this test), every one found again with its own
    values, none twice, absent keys absent."""
    N, L = pkg._native, LIB(pkg)
    cap_log2 = 14
    rows = int(load * (1 << cap_log2))
    rng = np.random.default_rng(7)
    words = 1 if n == 4 else 2
    keys = rng.integers(1, 1 << 62, size=(2 * rows, words), dtype=np.int64)
    if words == 2:
        keys |= np.int64(-(1 << 63))                      # 5x5 key words carry bit 63
    keys = np.unique(keys, axis=0)[: rows + 500]
    rng.shuffle(keys)
    present, absent = keys[:rows], keys[rows:]
    q = rng.standard_normal((rows, 4)).astype(np.float32)
    agent = pkg.BatchedQLearningAgent(10, capacity_log2=cap_log2, device=DEV, board_size=n)
    tk, tq = torch.from_numpy(present.copy()).to(DEV), torch.from_numpy(q).to(DEV)
    st = torch.zeros(1, dtype=torch.int32, device=DEV)
    assert L.q2048_table_import(agent.table.data_ptr(), cap_log2, tk.data_ptr(), tq.data_ptr(), rows, words,
                                st.data_ptr(), None) == 0
    assert int(st.item()) == 0 and agent.table_size() == rows
    k2, q2 = agent.export_rows()
    k2 = k2.reshape(len(q2), -1).view(np.int64)
    o1, o2 = np.lexsort(present.T[::-1]), np.lexsort(k2.T[::-1])
    assert np.array_equal(present[o1], k2[o2]) and np.array_equal(q[o1], q2[o2])
    # importing the same keys again overwrites in place: no second row for any key
    assert L.q2048_table_import(agent.table.data_ptr(), cap_log2, tk.data_ptr(), tq.data_ptr(), rows, words,
                                st.data_ptr(), None) == 0
    assert agent.table_size() == rows
    if n == 4:     # a 4x4 key is its board's 16 nibbles: look every key up through the board path
        def boards_of(kk):
            k = kk[:, 0].astype(np.uint64)
            return np.stack([((k >> np.uint64(4 * c)) & np.uint64(15)).astype(np.uint8) for c in range(16)], axis=1)
        got, found = agent.q_values(t8(boards_of(present)), return_found=True)
        assert bool(found.all()) and np.array_equal(got.cpu().numpy(), q)
        got, found = agent.q_values(t8(boards_of(absent)), return_found=True)
        assert not bool(found.any()) and float(got.abs().max()) == 0.0


def test_import_reports_rows_beyond_the_learning_probe_limit(pkg):
    """bulk imports place rows up to 2^14 slots down their probe sequence, the learning paths look 2^10
    slots far.  A table imported to load 0.99 (clusters thousands of slots long) holds such rows: the import says so
    (Q2048_STATUS_DEEP_ROW; no row is lost: TABLE_FULL stays clear and the count is exact), `q_values` (2^14) finds
    every row -- and the Python host refuses to load a checkpoint beyond load 0.9 in the first place."""
    N, L = pkg._native, LIB(pkg)
    cap_log2, rows = 14, int(0.99 * (1 << 14))
    rng = np.random.default_rng(11)
    keys = np.unique(rng.integers(1, 1 << 62, size=2 * rows, dtype=np.int64))[:rows]
    rng.shuffle(keys)
    q = rng.standard_normal((rows, 4)).astype(np.float32)
    agent = pkg.BatchedQLearningAgent(10, capacity_log2=cap_log2, device=DEV)
    tk, tq = torch.from_numpy(keys.copy()).to(DEV), torch.from_numpy(q).to(DEV)
    st = torch.zeros(1, dtype=torch.int32, device=DEV)
    assert L.q2048_table_import(agent.table.data_ptr(), cap_log2, tk.data_ptr(), tq.data_ptr(), rows, 1, st.data_ptr(), None) == 0
    code = int(st.item())
    assert code & N.STATUS_DEEP_ROW and not code & N.STATUS_TABLE_FULL and agent.table_size() == rows
    k = keys.astype(np.uint64)
    boards = np.stack([((k >> np.uint64(4 * c)) & np.uint64(15)).astype(np.uint8) for c in range(16)], axis=1)
    got, found = agent.q_values(t8(boards), return_found=True)
    assert bool(found.all()) and np.array_equal(got.cpu().numpy(), q)
    fresh = pkg.BatchedQLearningAgent(10, capacity_log2=cap_log2, device=DEV)
    with pytest.raises(ValueError, match="load factor would exceed 0.9"):
        fresh.import_rows(keys.astype(np.uint64), q)


def test_shared_table_pure_exploration_trajectories_exact(pkg, O):
    """eps = 1: actions come from the draws alone, so every board trajectory is independent of
    the (racy) shared table and must equal the oracle bit for bit at any batch size."""
    B, steps, seed, id0, lr = 20000, 150, 4, 55, 0.1
    env = pkg.BatchedGame2048Env(B, seed=seed, env_id0=id0, device=DEV)
    # gamma = 0: the TD target is the reward alone (Agent/main.py:42), so the value of an entry
    # depends only on the rewards applied to it, not on when its neighbours were updated
    agent = pkg.BatchedQLearningAgent(100, exploration_rate=1.0, learning_rate=lr,
                                      discount_factor=0.0, capacity_log2=23, seed=seed,
                                      env_id0=id0, device=DEV)
    agent.fused_rollout(env, steps)
    envs = O.envs_init(B, 4, seed, id0)
    oa = O.Agent(100, 4, lr, 0.0, 1.0)
    si, sf = np.zeros(O.ST_NI, np.int64), np.zeros(O.SF_NF)
    touched = []                                 # (state key, action, reward) of every update
    for t in range(steps):
        s = envs["board"][:, :16].copy()
        i1, f1, acts, rew, dn = O.rollout(envs, oa, 1, seed, id0, t, record=True)
        si += i1; sf += f1
        key = (s.astype(np.uint64) << (4 * np.arange(16, dtype=np.uint64))).sum(axis=1, dtype=np.uint64)
        touched.append(np.stack([key, acts[0].astype(np.uint64),
                                 rew[0].astype(np.float32).view(np.uint32).astype(np.uint64)], axis=1))
    assert np.array_equal(env.boards.cpu().numpy(), envs["board"][:, :16])
    assert_aux(env.aux_fields(), envs, "shared eps=1")
    st = agent.stats()
    assert st["steps"] == B * steps and st["episodes"] == si[O.ST_EPISODES]
    assert st["valid_moves"] == si[O.ST_VALID] and st["score_sum"] == si[O.ST_SCORE]
    assert st["explored"] == B * steps and st["drops"] == 0
    # the shared table holds exactly the rows of the reference's dict (every state touched by
    # update_q_value), whatever the order in which the lanes created them
    assert st["inserts"] == agent.table_size() == len(oa)
    keys, vals = oa.dump()
    q, found = agent.q_values(t8(keys), return_found=True)
    assert bool(found.cpu().numpy().all())
    # sequential (oracle) vs concurrent (device) update order differs only where lanes share an
    # entry: every (s, a) updated exactly once in the whole run holds lr * reward on both sides
    # (device vs the formula AND vs the oracle's float64 table), an entry updated k times holds a
    # value between the extremes of {0, its rewards} (each update moves it a fraction lr towards
    # one of them), and entries of touched rows that no update wrote are exactly 0
    d = agent.export_dict()
    assert len(d) == agent.table_size()
    tch = np.concatenate(touched)
    order = np.lexsort((tch[:, 1], tch[:, 0]))
    sp = tch[order, :2]                                       # (key, action), grouped
    rw = tch[order, 2].astype(np.uint32).view(np.float32).astype(np.float64)
    new = np.ones(len(sp), dtype=bool)
    new[1:] = (sp[1:] != sp[:-1]).any(axis=1)
    starts = np.flatnonzero(new)
    counts = np.diff(np.append(starts, len(sp)))
    ent_key, ent_act = sp[starts, 0], sp[starts, 1].astype(np.int64)

    def unpack(k):
        return np.stack([(k >> np.uint64(4 * c)) & np.uint64(15) for c in range(16)], axis=1)

    rows_k, row_of = np.unique(ent_key, return_inverse=True)
    q_rows = agent.q_values(t8(unpack(rows_k))).cpu().numpy().astype(np.float64)
    got = q_rows[row_of, ent_act]
    once = counts == 1
    assert once.sum() > 0.5 * len(starts)                     # most entries are touched once
    assert np.allclose(got[once], lr * rw[starts[once]], rtol=1e-5, atol=1e-6)
    okeys = (keys.astype(np.uint64) << (4 * np.arange(16, dtype=np.uint64))).sum(axis=1, dtype=np.uint64)
    osort = np.argsort(okeys)
    q_orc = vals[osort][np.searchsorted(okeys[osort], ent_key), ent_act]
    assert np.allclose(got[once], q_orc[once], rtol=1e-5, atol=1e-6)
    lo = np.minimum.reduceat(np.minimum(rw, 0.0), starts)
    hi = np.maximum.reduceat(np.maximum(rw, 0.0), starts)
    assert np.all(got >= lo - 1e-6) and np.all(got <= hi + 1e-6)
    touched_a = np.zeros((len(rows_k), 4), dtype=bool)
    touched_a[row_of, ent_act] = True
    assert np.all(q_rows[~touched_a] == 0.0)
    print(f"[q] shared eps=1: {int(once.sum())} of {len(starts)} entries touched once, all equal lr * reward")
    assert agent.check_status() == 0


def test_sharding_invariance(pkg):
    """Global env ids key the RNG: one 4096-env batch == two 2048-env shards (eps = 1)."""
    seed, steps = 21, 100
    shards = [pkg.shard_plan(4096, 2, r, base_env_id=1000) for r in range(2)]

    def run(B, id0):
        e = pkg.BatchedGame2048Env(B, seed=seed, env_id0=id0, device=DEV)
        a = pkg.BatchedQLearningAgent(100, exploration_rate=1.0, capacity_log2=20, seed=seed,
                                      env_id0=id0, device=DEV)
        a.fused_rollout(e, steps)
        return e.boards.cpu().numpy(), a.stats_i.cpu().numpy(), a.stats_f.cpu().numpy()

    full_b, full_i, full_f = run(4096, 1000)
    parts = [run(s.num_envs, s.env_id0) for s in shards]
    assert np.array_equal(np.concatenate([p[0] for p in parts]), full_b)
    mi, mf = pkg.dist.merge_stats_numpy([p[1] for p in parts], [p[2] for p in parts])
    skip = {pkg._native.ST_INSERTS, pkg._native.ST_CAS_RETRY}   # table replicas differ
    for k in range(pkg._native.NSTAT_I):
        if k not in skip:
            assert mi[k] == full_i[k], k
    assert np.allclose(mf, full_f, rtol=1e-9)


def test_reference_surface_adapters(pkg, O):
    """The reference loop body (Agent/main.py:80-109) runs unchanged on the adapters, and what it
    computes is the reference's run: every board, action, reward and done flag, every Q row the
    loop reads at :96 (returned lazily, read back thousands of steps later) and the final dict
    equal the oracle's B = 1 loop under the same draws."""
    seed = 3
    env = pkg.Game2048_env(device=DEV, seed=seed)
    num_episodes = 3
    agent = pkg.QLearningAgent(num_episodes, action_space=env.action_space.n, device=DEV, seed=seed)
    envs = O.envs_init(1, 4, seed, 0)
    # the loop resets the env before its first episode too (Agent/main.py:81): the constructor's
    # game is never played; every later reset is the one the oracle's rollout makes on `done`
    first = np.asarray(O.draws(seed, 0, 1, O.STREAM_RESET), dtype=np.uint32)
    O.lib().orc_env_reset(envs.ctypes.data, O._u32(first))
    envs["episode"][0] = 1
    oa = O.Agent(num_episodes, 4)                      # the reference's defaults (Agent/main.py:15)
    t, rows, want_rows = 0, [], []
    for episode in range(num_episodes):
        state = env.reset()
        assert state.shape == (4, 4) and state.dtype == np.int64
        assert pkg.raw_to_boards(state).reshape(-1).tolist() == envs["board"][0, :16].tolist()
        state = tuple(map(tuple, state))
        done, total_reward, n = False, 0, 0
        while not done and n < 3000:
            action = agent.choose_action(state)
            next_state, reward, done, info = env.step(action)
            next_state = tuple(map(tuple, next_state))
            q_values = agent.q_table[state]            # :96 -- before the update of :99 is queued
            want_rows.append(oa.q(envs["board"][0, :16]))
            si, sf, oact, orew, odn = O.rollout(envs, oa, 1, seed, 0, t, record=True)
            assert (action, bool(done)) == (int(oact[0, 0]), bool(odn[0, 0])), t
            if done:                                   # env.score of the finished episode (:104)
                assert env.score == int(si[O.ST_SCORE]) > 0
            assert abs(reward - float(np.float32(orew[0, 0]))) <= float(ulp32(orew[0, 0])), t
            if not done:                               # (on done the oracle has already reset)
                assert pkg.raw_to_boards(np.array(next_state)).reshape(-1).tolist() == envs["board"][0, :16].tolist(), t
            agent.update_q_value(state, action, reward, next_state, done)
            rows.append(q_values)                      # not looked at until the run is over
            state = next_state
            total_reward += reward
            n += 1
            t += 1
        assert done and isinstance(reward, float) and isinstance(info, int) and isinstance(action, int)
        assert q_values.shape == (4,) and len(q_values) == 4 and info == np.max(env.game.board)
        agent.decay_exploration(episode)
        oa.decay_exploration(episode)
    assert agent.epsilon == oa.epsilon == 0.01 and len(agent.q_table) == len(oa) > 10
    assert t > 200 and len(rows) == t
    got = np.stack([np.asarray(r) for r in rows])
    assert got.dtype == np.float64 and np.allclose(got, np.stack(want_rows), rtol=1e-5, atol=1e-6)
    keys, vals = oa.dump()
    final = np.stack([np.asarray(agent.q_table[tuple(map(tuple, pkg.boards_to_raw(k)))]) for k in keys[:200]])
    assert np.allclose(final, vals[:200], rtol=1e-5, atol=1e-6)
    assert str(rows[0]).startswith("[") and float(rows[-1][0]) == got[-1, 0] and rows[5].tolist() == got[5].tolist()


def test_adapters_follow_the_table_policy(pkg, O):
    """The one-env drop-ins on a table far too small for the run (2^9 slots): the reference's loop body goes on
    unchanged -- the table closes its key set when half of it is in use (checked before every `update_q_value`, as the
    batched agent does before every launch), unknown states are played on visit rows, and every action, reward and
    done flag equals the oracle's sequential agent whose dict stops taking keys at the same moment."""
    import warnings

    seed, eps, steps = 9, 0.1, 1500
    env = pkg.Game2048_env(device=DEV, seed=seed)
    agent = pkg.QLearningAgent(100, action_space=4, exploration_rate=eps, discount_factor=0.99, capacity_log2=9,
                               device=DEV, seed=seed)
    envs = O.envs_init(1, 4, seed, 0)
    first = np.asarray(O.draws(seed, 0, 1, O.STREAM_RESET), dtype=np.uint32)
    O.lib().orc_env_reset(envs.ctypes.data, O._u32(first))
    envs["episode"][0] = 1
    oa = O.Agent(100, 4, 0.1, 0.99, eps)
    state, t, frozen_at = tuple(map(tuple, env.reset())), 0, None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        while t < steps:
            action = agent.choose_action(state)
            next_state, reward, done, _ = env.step(action)
            next_state = tuple(map(tuple, next_state))
            if frozen_at is None and len(oa) >= 0.5 * (1 << 9):
                oa.freeze()
                frozen_at = t
            _, _, oact, orew, odn = O.rollout(envs, oa, 1, seed, 0, t, record=True)
            assert (action, bool(done)) == (int(oact[0, 0]), bool(odn[0, 0])), (t, frozen_at)
            assert abs(reward - float(np.float32(orew[0, 0]))) <= float(ulp32(orew[0, 0])), t
            agent.update_q_value(state, action, reward, next_state, done)
            state = tuple(map(tuple, env.reset())) if done else next_state
            t += 1
    b = agent._b
    assert b.frozen and frozen_at is not None and frozen_at < steps // 2 and b.frozen_at["rows"] == len(oa) == len(agent.q_table)
    assert b.stats()["drops"] == oa.drops > 100 and b.check_status() == 0
    keys, vals = oa.dump()
    got = np.stack([np.asarray(agent.q_table[tuple(map(tuple, pkg.boards_to_raw(k)))]) for k in keys])
    assert np.allclose(got, vals, rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------------------------------------
# 5x5 boards (BASELINE configs[4]).  The reference hard-codes 4x4; the oracle's n-generic
# restatement (pinned to the reference at n = 4) is the checker at n = 5.
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B", [1, 255, 256, 1000])
def test_5x5_env_init_and_rollout_match_oracle(pkg, O, B):
    """25-byte boards staged through LDS (ragged last block), step / reset(done) vs the oracle."""
    steps, seed, id0 = 300, 61, 4_000_000_000
    rng = np.random.default_rng(B)
    actions = np.where(rng.random((steps, B)) < 0.6, rng.integers(0, 2, size=(steps, B)),
                       rng.integers(0, 4, size=(steps, B))).astype(np.uint8)
    actions[:, : max(1, B // 50)] = 2
    envs = O.envs_init(B, 5, seed, id0)
    env = pkg.BatchedGame2048Env(B, board_size=5, seed=seed, env_id0=id0, device=DEV)
    assert env.boards.shape == (B, 25)
    assert np.array_equal(env.boards.cpu().numpy(), envs["board"][:, :25])
    si, sf, _, rew, dn = O.rollout(envs, None, steps, seed, id0, 0, actions=actions, record=True)
    acts = torch.from_numpy(actions).to(DEV)
    drew, ddn = [], []
    for t in range(steps):
        _, r, d, mt = env.step(acts[t])
        drew.append(r.clone()); ddn.append(d.clone())
        env.reset(d)
    ddn = torch.stack(ddn).cpu().numpy().astype(np.uint8)
    assert np.array_equal(ddn, dn)
    assert np.array_equal(env.boards.cpu().numpy(), envs["board"][:, :25])
    assert_rewards(torch.stack(drew).cpu().numpy(), rew, "5x5 rollout")
    assert_aux(env.aux_fields(), envs, "5x5 rollout")
    assert env.check_status() == 0


def test_5x5_fused_rollout_matches_oracle_independent_lanes(pkg, O):
    B, steps, seed, id0, eps, lr, gamma = 150, 400, 23, 987654, 0.2, 0.1, 0.99
    envs = O.envs_init(B, 5, seed, id0)
    agents = [O.Agent(100, 4, lr, gamma, eps, n=5) for _ in range(B)]
    tot_i = np.zeros(O.ST_NI, np.int64)
    for i in range(B):
        si, sf = O.rollout(envs[i:i + 1], agents[i], steps, seed, id0 + i, 0)
        tot_i += si
    env = pkg.BatchedGame2048Env(B, board_size=5, seed=seed, env_id0=id0, device=DEV)
    agent = pkg.BatchedQLearningAgent(100, learning_rate=lr, discount_factor=gamma,
                                      exploration_rate=eps, capacity_log2=18, seed=seed, env_id0=id0,
                                      device=DEV, independent=True, board_size=5)
    for k in (1, 99, 300):                       # split launches: state carried in HBM in between
        agent.fused_rollout(env, k)
    assert np.array_equal(env.boards.cpu().numpy(), envs["board"][:, :25])
    assert_aux(env.aux_fields(), envs, "5x5 fused")
    worst = 0.0
    for i, oa in enumerate(agents):
        keys, vals = oa.dump()
        got = agent.q_values(t8(keys), env_id=id0 + i).cpu().numpy()
        assert np.allclose(got, vals, rtol=1e-5, atol=1e-6), i
        worst = max(worst, float(np.max(np.abs(got - vals) / (np.abs(vals) + 1e-1))))
    print(f"[q] 5x5 fused: worst relative Q error {worst:.2e}")
    # agent.q_table[state] (Agent/main.py:16,96) with a 5x5 state: rows of env id0 (lane 0)
    k0, v0 = agents[0].dump()
    for j in (0, len(k0) // 2, len(k0) - 1):
        state = tuple(map(tuple, pkg.boards_to_raw(k0[j])))
        assert len(state) == 5 and np.allclose(agent.q_table[state], v0[j], rtol=1e-5, atol=1e-6)
    st = agent.stats()
    assert st["steps"] == B * steps and st["episodes"] == tot_i[O.ST_EPISODES]
    assert st["valid_moves"] == tot_i[O.ST_VALID] and st["score_sum"] == tot_i[O.ST_SCORE]
    assert st["inserts"] == agent.table_size() == sum(len(oa) for oa in agents)
    assert st["drops"] == 0 and agent.check_status() == 0


def test_5x5_fused_equals_unfused(pkg):
    B, steps, seed, id0 = 200, 80, 5, 31

    def mk():
        e = pkg.BatchedGame2048Env(B, board_size=5, seed=seed, env_id0=id0, device=DEV)
        a = pkg.BatchedQLearningAgent(100, exploration_rate=0.3, capacity_log2=17, seed=seed,
                                      env_id0=id0, device=DEV, independent=True, board_size=5)
        return e, a

    e1, a1 = mk(); a1.fused_rollout(e1, steps)
    e2, a2 = mk(); _unfused_loop(pkg, e2, a2, steps)
    assert torch.equal(e1.boards, e2.boards)
    k1, q1 = a1.export_rows(); k2, q2 = a2.export_rows()
    o1 = np.lexsort((k1[:, 1], k1[:, 0])); o2 = np.lexsort((k2[:, 1], k2[:, 0]))
    assert np.array_equal(k1[o1], k2[o2]) and np.array_equal(q1[o1], q2[o2])


def test_5x5_shared_table_pure_exploration(pkg, O):
    """eps = 1 on a shared table: trajectories and the table's key set equal the oracle's."""
    B, steps, seed, id0 = 5000, 120, 77, 1
    env = pkg.BatchedGame2048Env(B, board_size=5, seed=seed, env_id0=id0, device=DEV)
    agent = pkg.BatchedQLearningAgent(100, exploration_rate=1.0, capacity_log2=21, seed=seed,
                                      env_id0=id0, device=DEV, board_size=5)
    agent.fused_rollout(env, steps)
    envs = O.envs_init(B, 5, seed, id0)
    oa = O.Agent(100, 4, 0.1, 0.9, 1.0, n=5)
    si, sf = O.rollout(envs, oa, steps, seed, id0, 0)
    assert np.array_equal(env.boards.cpu().numpy(), envs["board"][:, :25])
    st = agent.stats()
    assert st["inserts"] == agent.table_size() == len(oa)
    assert st["valid_moves"] == si[O.ST_VALID] and st["episodes"] == si[O.ST_EPISODES]
    d = agent.export_dict()
    keys, vals = oa.dump()
    want = {tuple(tuple(int(v) for v in r) for r in pkg.boards_to_raw(k)) for k in keys}
    assert set(d.keys()) == want
    assert agent.check_status() == 0


# ---------------------------------------------------------------------------------------------
# full size, table limits, checkpoint
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,launches,S,cap", [(1 << 20, 3, 16, 27), (1 << 23, 2, 8, 28)])
def test_full_size_1m_boards_properties(pkg, O, B, launches, S, cap):
    """BASELINE configs[2] size (1,048,576 boards, shared hash table) and the whole configs[3]
    batch (8,388,608 boards) on one GPU.  eps = 1 makes every board trajectory a pure function of
    (seed, global env id, step), so sampled lanes are checked bit-exactly against the oracle and
    the whole batch through size-independent invariants."""
    seed, id0 = 2024, 7
    env = pkg.BatchedGame2048Env(B, seed=seed, env_id0=id0, device=DEV)
    agent = pkg.BatchedQLearningAgent(1000, exploration_rate=1.0, discount_factor=0.99,
                                      capacity_log2=cap, seed=seed, env_id0=id0, device=DEV)
    for _ in range(launches):
        agent.fused_rollout(env, S)
    steps = launches * S
    boards = env.boards.cpu().numpy()
    aux = env.aux_fields()
    rng = np.random.default_rng(0)
    sample = np.unique(np.concatenate([[0, 1, 63, 64, 255, 256, B - 257, B - 2, B - 1],
                                       rng.integers(0, B, size=1500)]))
    for i in sample.tolist():
        envs = O.envs_init(1, 4, seed, id0 + i)
        O.rollout(envs, O.Agent(10, 4, 0.1, 0.99, 1.0), steps, seed, id0 + i, 0)
        assert boards[i].tolist() == envs["board"][0, :16].tolist(), i
        assert aux["score"][i] == envs["score"][0] and aux["episode"][i] == envs["episode"][0], i
        assert aux["cons_count"][i] == envs["consecutive_count"][0], i
    st = agent.stats()
    assert st["steps"] == B * steps and st["explored"] == B * steps
    assert st["episodes"] == int(aux["episode"].astype(np.int64).sum())      # every reset is counted
    assert st["inserts"] == agent.table_size() and st["drops"] == 0
    assert agent.check_status() == 0 and env.check_status() == 0
    assert boards.max() <= 15 and (boards > 0).sum(axis=1).min() >= 2        # legal boards only
    assert sum(st["max_tile_hist"].values()) == st["episodes"]
    # determinism: a second run of the same job gives the same boards and statistics
    del agent
    import gc
    gc.collect(); torch.cuda.empty_cache()
    env2 = pkg.BatchedGame2048Env(B, seed=seed, env_id0=id0, device=DEV)
    agent2 = pkg.BatchedQLearningAgent(1000, exploration_rate=1.0, discount_factor=0.99,
                                       capacity_log2=cap, seed=seed, env_id0=id0, device=DEV)
    agent2.fused_rollout(env2, steps)            # one launch instead of three
    assert torch.equal(env.boards, env2.boards) and torch.equal(env.aux, env2.aux)
    st2 = agent2.stats()
    for k in ("steps", "episodes", "valid_moves", "score_sum", "inserts", "max_tile_hist"):
        assert st[k] == st2[k], k


@pytest.mark.parametrize("B", [1 << 20, (1 << 20) + 77])
def test_5x5_full_size_1m_boards(pkg, O, B):
    """BASELINE configs[4] at its stated size: 1,048,576 5x5 boards through the fused kernel on a
    shared table (and the same + 77 lanes, so that the last block of the 25-byte board stream is
    ragged).  eps = 1: sampled lanes bit-exact against the oracle -- the first and last lanes of
    the stream, block edges and 1500 random ones, which exercises the 64-bit `base * 25`
    addressing of the unpadded stream end to end -- and the whole batch by invariants."""
    seed, id0, launches, S, cap = 909, 3, 3, 16, 27
    env = pkg.BatchedGame2048Env(B, board_size=5, seed=seed, env_id0=id0, device=DEV)
    agent = pkg.BatchedQLearningAgent(1000, exploration_rate=1.0, discount_factor=0.99,
                                      capacity_log2=cap, seed=seed, env_id0=id0, device=DEV,
                                      board_size=5)
    for _ in range(launches):
        agent.fused_rollout(env, S)
    steps = launches * S
    boards = env.boards.cpu().numpy()
    assert boards.shape == (B, 25)
    aux = env.aux_fields()
    rng = np.random.default_rng(5)
    last_block = (B - 1) // 256 * 256
    sample = np.unique(np.concatenate([[0, 1, 63, 64, 255, 256, 257, last_block - 1, last_block,
                                        B - 2, B - 1], rng.integers(0, B, size=1500)]))
    for i in sample.tolist():
        envs = O.envs_init(1, 5, seed, id0 + i)
        O.rollout(envs, O.Agent(10, 4, 0.1, 0.99, 1.0, n=5), steps, seed, id0 + i, 0)
        assert boards[i].tolist() == envs["board"][0, :25].tolist(), i
        assert aux["score"][i] == envs["score"][0] and aux["episode"][i] == envs["episode"][0], i
        assert aux["cons_count"][i] == envs["consecutive_count"][0], i
        assert aux["prev_max"][i] == envs["previous_max_log2"][0], i
    st = agent.stats()
    assert st["steps"] == B * steps and st["explored"] == B * steps
    assert st["episodes"] == int(aux["episode"].astype(np.int64).sum())
    assert st["inserts"] == agent.table_size() and st["drops"] == 0
    assert agent.check_status() == 0 and env.check_status() == 0
    assert pkg._native.claim_timeouts() == 0
    assert boards.max() <= 31 and (boards > 0).sum(axis=1).min() >= 2
    # a second run in one launch instead of three: same boards, same statistics
    del agent
    import gc
    gc.collect(); torch.cuda.empty_cache()
    env2 = pkg.BatchedGame2048Env(B, board_size=5, seed=seed, env_id0=id0, device=DEV)
    agent2 = pkg.BatchedQLearningAgent(1000, exploration_rate=1.0, discount_factor=0.99,
                                       capacity_log2=cap, seed=seed, env_id0=id0, device=DEV,
                                       board_size=5)
    agent2.fused_rollout(env2, steps)
    assert torch.equal(env.boards, env2.boards) and torch.equal(env.aux, env2.aux)
    st2 = agent2.stats()
    for k in ("steps", "episodes", "valid_moves", "score_sum", "inserts", "max_tile_hist"):
        assert st[k] == st2[k], k


@pytest.mark.parametrize("n,cap", [(4, 27), (5, 27)])
def test_full_size_1m_lanes_q_dependent_actions(pkg, O, n, cap):
    """1,048,576 lanes with eps = 0.2, so that 80 % of the actions are argmax over Q rows the
    lane itself learnt (Agent/main.py:38): Q2048_FLAG_INDEPENDENT gives every env private rows,
    which makes each lane a pure function of its own global id at any batch size.  >= 1000 sampled
    lanes are compared with one oracle agent each: boards and aux bit-exact, EVERY Q row of the
    lane within rtol 1e-5 (the north-star tolerance)."""
    B, seed, id0, eps, lr, gamma, launches, S = 1 << 20, 31, 11, 0.2, 0.1, 0.99, 3, 16
    env = pkg.BatchedGame2048Env(B, board_size=n, seed=seed, env_id0=id0, device=DEV)
    agent = pkg.BatchedQLearningAgent(1000, learning_rate=lr, discount_factor=gamma,
                                      exploration_rate=eps, capacity_log2=cap, seed=seed,
                                      env_id0=id0, device=DEV, independent=True, board_size=n)
    for _ in range(launches):
        agent.fused_rollout(env, S)
    steps = launches * S
    boards = env.boards.cpu().numpy()
    aux = env.aux_fields()
    rng = np.random.default_rng(n)
    sample = np.unique(np.concatenate([[0, 1, 63, 64, 255, 256, B - 257, B - 2, B - 1],
                                       rng.integers(0, B, size=1100)]))
    assert len(sample) >= 1000
    worst, rows, greedy = 0.0, 0, 0
    for i in sample.tolist():
        envs = O.envs_init(1, n, seed, id0 + i)
        oa = O.Agent(1000, 4, lr, gamma, eps, n=n)
        si, _ = O.rollout(envs, oa, steps, seed, id0 + i, 0)
        greedy += steps - int(si[O.ST_EXPLORE])
        assert boards[i].tolist() == envs["board"][0, :n * n].tolist(), i
        assert aux["score"][i] == envs["score"][0] and aux["episode"][i] == envs["episode"][0], i
        assert aux["cons_count"][i] == envs["consecutive_count"][0], i
        assert aux["cons_action"][i] == envs["consecutive_action"][0] & 0xFF, i
        keys, vals = oa.dump()
        got, found = agent.q_values(t8(keys), env_id=id0 + i, return_found=True)
        assert bool(found.all()), i                      # every row of the reference dict exists
        got = got.cpu().numpy()
        assert np.allclose(got, vals, rtol=1e-5, atol=1e-6), i
        worst = max(worst, float(np.max(np.abs(got - vals) / (np.abs(vals) + 1e-1))))
        rows += len(keys)
    assert greedy > 0.7 * len(sample) * steps            # the argmax path was the one exercised
    print(f"[q] {n}x{n} 1M lanes eps={eps}: {len(sample)} lanes, {rows} rows, worst relative Q error {worst:.2e}")
    st = agent.stats()
    assert st["steps"] == B * steps and st["drops"] == 0 and st["cas_retries"] == 0
    assert st["inserts"] == agent.table_size()
    assert agent.check_status() == 0 and pkg._native.claim_timeouts() == 0


@pytest.mark.parametrize("n,cap", [(4, 27), (5, 27)])
def test_full_size_1m_lanes_closed_key_set(pkg, O, n, cap):
    """BASELINE configs[2] / [4] at full size with the key set CLOSED after the first launch: 1,048,576 lanes with
    private rows, eps = 0.2, one 16-step launch of ordinary learning, then three 16-step launches carrying
    Q2048_FLAG_NO_NEW_ROWS (visit rows cross the launch boundaries in the row cache).  >= 600 sampled lanes against
    one oracle agent each (`freeze()` after 16 steps): boards and aux bit-exact, every row of the lane's closed key
    set within 1e-5, and -- over the whole batch -- the table is exactly as large as it was when it closed."""
    B, seed, id0, eps, lr, gamma, S = 1 << 20, 33, 5, 0.2, 0.1, 0.99, 16
    env = pkg.BatchedGame2048Env(B, board_size=n, seed=seed, env_id0=id0, device=DEV)
    agent = pkg.BatchedQLearningAgent(1000, learning_rate=lr, discount_factor=gamma, exploration_rate=eps,
                                      capacity_log2=cap, seed=seed, env_id0=id0, device=DEV, independent=True,
                                      board_size=n, freeze_load=None)
    agent.fused_rollout(env, S)
    rows1 = agent.table_size()
    agent.frozen = True
    for _ in range(3):
        agent.fused_rollout(env, S)
    boards, aux = env.boards.cpu().numpy(), env.aux_fields()
    rng = np.random.default_rng(100 + n)
    sample = np.unique(np.concatenate([[0, 1, 63, 64, 511, 512, B - 513, B - 2, B - 1], rng.integers(0, B, size=650)]))
    assert len(sample) >= 600
    worst, drops = 0.0, 0
    for i in sample.tolist():
        envs = O.envs_init(1, n, seed, id0 + i)
        oa = O.Agent(1000, 4, lr, gamma, eps, n=n)
        O.rollout(envs, oa, S, seed, id0 + i, 0)
        size1 = len(oa)
        oa.freeze()
        for k in range(3):
            O.rollout(envs, oa, S, seed, id0 + i, S * (k + 1))
        assert len(oa) == size1
        drops += oa.drops
        assert boards[i].tolist() == envs["board"][0, :n * n].tolist(), i
        assert aux["score"][i] == envs["score"][0] and aux["episode"][i] == envs["episode"][0], i
        assert aux["cons_count"][i] == envs["consecutive_count"][0], i
        keys, vals = oa.dump()
        got, found = agent.q_values(t8(keys), env_id=id0 + i, return_found=True)
        assert bool(found.all()), i
        got = got.cpu().numpy()
        assert np.allclose(got, vals, rtol=1e-5, atol=1e-6), i
        worst = max(worst, float(np.max(np.abs(got - vals) / (np.abs(vals) + 1e-1))))
    st = agent.stats()
    print(f"[closed key set {n}x{n} 1M lanes] {len(sample)} lanes, {drops} of {len(sample) * 3 * S} sampled updates dropped, "
          f"worst relative Q error {worst:.2e}; batch: {rows1} rows, {st['drops']} drops")
    assert st["steps"] == 4 * B * S and st["inserts"] == rows1 == agent.table_size() and st["drops"] > B
    assert agent.check_status() == 0 and pkg._native.claim_timeouts() == 0


@pytest.mark.parametrize("n", [4, 5])
def test_full_size_1m_lanes_four_call_api(pkg, O, n):
    """The reference's four calls (choose_action, step, update_q_value, reset(done); no board copy, row
    cache) at 1,048,576 lanes with private rows (Q2048_FLAG_INDEPENDENT) and eps = 0.2: 400 sampled lanes
    against one oracle agent each -- boards and aux bit-exact, every Q row of the lane within rtol 1e-5.
    (`k_q_update` issues the claim of s' before the TD write of s and reads its answer afterwards: this is
    that kernel at the bench's size, 0.65 claims per lane-step in flight at once.)"""
    B, seed, id0, eps, lr, gamma, steps = 1 << 20, 77, 5, 0.2, 0.1, 0.99, 40
    env = pkg.BatchedGame2048Env(B, board_size=n, seed=seed, env_id0=id0, device=DEV)
    agent = pkg.BatchedQLearningAgent(1000, learning_rate=lr, discount_factor=gamma, exploration_rate=eps,
                                      capacity_log2=27, seed=seed, env_id0=id0, device=DEV, independent=True,
                                      board_size=n)
    s = env.boards
    for _ in range(steps):
        a = agent.choose_action(s)
        s2, r, d, _ = env.step(a)                       # the other board buffer: `s` stays intact
        agent.update_q_value(s, a, r, s2, d)
        s = env.reset(d)
    boards = env.boards.cpu().numpy()
    aux = env.aux_fields()
    rng = np.random.default_rng(9)
    sample = np.unique(np.concatenate([[0, 63, 64, 1023, 1024, B - 1025, B - 1], rng.integers(0, B, size=400)]))
    worst, rows = 0.0, 0
    for i in sample.tolist():
        envs = O.envs_init(1, n, seed, id0 + i)
        oa = O.Agent(1000, 4, lr, gamma, eps, n=n)
        O.rollout(envs, oa, steps, seed, id0 + i, 0)
        assert boards[i].tolist() == envs["board"][0, :n * n].tolist(), i
        assert aux["score"][i] == envs["score"][0] and aux["episode"][i] == envs["episode"][0], i
        assert aux["cons_count"][i] == envs["consecutive_count"][0], i
        keys, vals = oa.dump()
        got, found = agent.q_values(t8(keys), env_id=id0 + i, return_found=True)
        assert bool(found.all()), i
        got = got.cpu().numpy()
        assert np.allclose(got, vals, rtol=1e-5, atol=1e-6), i
        worst = max(worst, float(np.max(np.abs(got - vals) / (np.abs(vals) + 1e-1))))
        rows += len(keys)
    print(f"[q] 4-call API, {n}x{n}, 1M lanes: {len(sample)} lanes, {rows} rows, worst relative Q error {worst:.2e}")
    st = agent.stats()
    assert st["drops"] == 0 and st["inserts"] == agent.table_size()
    assert agent.check_status() == 0 and pkg._native.claim_timeouts() == 0


def test_table_full_drops_are_counted_not_raised(pkg, O):
    """A table that is too small: updates are dropped and counted, the status word says so,
    nothing raises, and the env trajectories are untouched."""
    B, steps, seed = 4096, 30, 5
    env = pkg.BatchedGame2048Env(B, seed=seed, device=DEV)
    agent = pkg.BatchedQLearningAgent(100, exploration_rate=1.0, capacity_log2=10, seed=seed, device=DEV)
    agent.fused_rollout(env, steps)
    st = agent.stats()
    assert st["drops"] > 0 and st["inserts"] <= 1 << 10 and st["inserts"] == agent.table_size()
    assert agent.check_status() & pkg._native.STATUS_TABLE_FULL
    envs = O.envs_init(B, 4, seed, 0)
    O.rollout(envs, O.Agent(100, 4, 0.1, 0.9, 1.0), steps, seed, 0, 0)
    assert np.array_equal(env.boards.cpu().numpy(), envs["board"][:, :16])


@pytest.mark.parametrize("n,freeze", [(4, 0.5), (5, 0.5), (4, None), (5, None)])
def test_rollout_on_a_full_table_stays_bounded(pkg, n, freeze):
    """A FIXED table of 2^22 slots under a run that wants far more rows than it has.
    freeze_load = 0.5 (the default policy, SURVEY 7.3 "stop inserting and count drops"): the table closes its key set
    within one launch of load 0.5; from then on the row count is constant, drops are counted, TABLE_FULL is NOT
    raised, and a launch costs at most 3 x a launch on the young table (the ratio is printed; round 5 allowed 200 x
    on a table driven to load 1.0).
    freeze_load = None (the table is left to fill up): ADVICE r4's bound -- the learning paths probe at most 2^10
    slots, so a launch on the FULL table stays within a (large) multiple of the young one, drops are counted, the
    status word says TABLE_FULL, and every row that was created is still found."""
    import time
    import warnings

    B, S, cap = 1 << 16, 8, (22 if freeze is not None else 20)
    env = pkg.BatchedGame2048Env(B, board_size=n, seed=12, device=DEV)
    agent = pkg.BatchedQLearningAgent(100, exploration_rate=1.0, capacity_log2=cap, seed=12, device=DEV, board_size=n,
                                      freeze_load=freeze)

    def launch():
        sync()
        t0 = time.perf_counter()
        agent.fused_rollout(env, S)
        sync()
        return time.perf_counter() - t0

    launch()
    young = min(launch() for _ in range(3))
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        for _ in range(40 if freeze is None else 60):     # ~0.6-0.9 rows per env-step
            agent.fused_rollout(env, S)
    st = agent.stats()
    rows = agent.table_size()
    load = rows / (1 << cap)
    full = sorted(launch() for _ in range(5))[2]          # (the median: one slow launch on a busy host is not the table)
    print(f"[full table {n}x{n}, freeze_load {freeze}] load {load:.3f}: {full * 1e3:.2f} ms per {S}-step launch "
          f"against {young * 1e3:.2f} ms on the young table = {full / young:.2f} x")
    assert st["drops"] > 0 and rows == st["inserts"]
    if freeze is None:
        assert rows > 0.9 * (1 << cap) and not agent.frozen
        assert agent.check_status() & pkg._native.STATUS_TABLE_FULL
        assert full < max(200 * young, 0.5), (full, young)
    else:
        assert agent.frozen and agent.frozen_at["rows"] == rows                     # nothing was created since
        assert freeze <= load <= freeze + 2.0 * B * S / (1 << cap), load           # within one launch of the limit
        assert agent.check_status() == 0                                            # the caller's policy: no TABLE_FULL
        assert sum("takes no new rows" in str(w.message) for w in caught) == 1      # said once
        assert agent.table_size() == rows and agent.stats()["inserts"] == rows     # the five launches above: none
        assert full < 3.0 * young, (full, young)
        assert agent.verify_table()["rows"] == rows
    k, q = agent.export_rows()                            # every created row is in the table, once
    assert len(q) == rows and len(np.unique(k.reshape(len(q), -1), axis=0)) == rows
    assert pkg._native.claim_timeouts(LIB(pkg)) == 0


@pytest.mark.parametrize("n,path,cache,eps,k1", [(4, "fused", True, 0.3, 50), (5, "fused", True, 0.3, 50),
                                                  (4, "four_call", True, 0.3, 50), (5, "four_call", True, 0.3, 50),
                                                  (4, "strict", True, 0.3, 50), (4, "fused", False, 0.3, 50),
                                                  (4, "four_call", False, 0.3, 50), (4, "fused", True, 0.0, 0),
                                                  (5, "fused", True, 0.02, 8), (4, "mixed", True, 0.3, 50),
                                                  (5, "mixed", True, 0.0, 0), (4, "det", True, 0.0, 0)])
def test_closed_key_set_private_rows_match_oracle(pkg, O, n, path, cache, eps, k1):
    """Q2048_FLAG_NO_NEW_ROWS with private rows (every env = one reference agent whose dict stops taking keys after
    k1 steps: the oracle's `freeze()`).  The actions depend on the rows, on absent states reading as the zero row the
    defaultdict would have created (Agent/main.py:16,38,41), AND on the env's VISIT ROW: while an env stays in a state
    without a row its dropped updates land in that fresh row, so an invalid action 0 is followed by action 1 (not by
    action 0 again until the >100-repeats rule ends the episode).  Boards and aux bit-exact, every row of the closed
    key set within 1e-5, the key set itself unchanged, drops == the oracle's count of updates whose state has no row,
    no TABLE_FULL.  Through the fused kernel (split launches: the visit row crosses the cut in the row cache; also with
    compare-and-swap TD writes) and the 4-call API; `cache` False: no row cache, a visit row ends with the call (the
    oracle's ENV_NEW_VISITS).  eps = 0 on an EMPTY closed table: pure greedy play on visit rows alone.
    "det": the deterministic step (with private rows the two-phase semantic IS the sequential one); "mixed": fused
    rollout, deterministic step and 4-call API in turn -- a visit row crosses every one of the boundaries in the row
    cache, whichever path wrote the record."""
    B, k2, seed, id0, lr, gamma = 96, 250, 23, 7000, 0.1, 0.99
    env = pkg.BatchedGame2048Env(B, board_size=n, seed=seed, env_id0=id0, device=DEV)
    agent = pkg.BatchedQLearningAgent(100, learning_rate=lr, discount_factor=gamma, exploration_rate=eps,
                                      capacity_log2=17, seed=seed, env_id0=id0, device=DEV, independent=True,
                                      board_size=n, freeze_load=None, strict_td=path == "strict", row_cache=cache)
    calls = []                                            # the device's calls, for the oracle to repeat

    def run(k):
        if k == 0:
            return
        if path == "four_call":
            _unfused_loop(pkg, env, agent, k)
            calls.extend([1] * k)
        elif path == "det":
            for part in (k // 3, k - k // 3):
                if part:
                    agent.deterministic_rollout(env, part)
                    calls.append(part)
        elif path == "mixed":
            left, turn = k, 0
            while left > 0:
                part = min(left, (17, 9, 5, 1, 23, 2)[turn % 6])
                if turn % 3 == 0:
                    agent.fused_rollout(env, part)
                    calls.append(part)
                elif turn % 3 == 1:
                    agent.deterministic_rollout(env, part)
                    calls.append(part)
                else:
                    _unfused_loop(pkg, env, agent, part)
                    calls.extend([1] * part)
                left, turn = left - part, turn + 1
        else:
            for part in (k // 3, k - k // 3):             # (split launches: the flag travels with each)
                if part:
                    agent.fused_rollout(env, part)
                    calls.append(part)

    run(k1)
    n1 = len(calls)
    rows1 = agent.table_size()
    agent.frozen = True                                   # the key set is closed by hand (the policy has its own test)
    run(k2)
    envs = O.envs_init(B, n, seed, id0)
    oflags = 0 if cache else O.ENV_NEW_VISITS
    drops, rows, worst, stalled = 0, 0, 0.0, 0
    for i in range(B):
        oa = O.Agent(100, 4, lr, gamma, eps, n=n)
        ctr = 0
        for c, steps in enumerate(calls):
            if c == n1:
                size1 = len(oa)
                oa.freeze()
            O.rollout(envs[i:i + 1], oa, steps, seed, id0 + i, ctr, env_flags=oflags)
            ctr += steps
        if n1 == 0:
            size1 = 0
        assert len(oa) == size1
        drops += oa.drops
        rows += size1
        keys, vals = oa.dump()
        if len(keys):
            got, found = agent.q_values(t8(keys), env_id=id0 + i, return_found=True)
            assert bool(found.all()), i
            got = got.cpu().numpy()
            assert np.allclose(got, vals, rtol=1e-5, atol=1e-6), i
            worst = max(worst, float(np.max(np.abs(got - vals) / (np.abs(vals) + 1e-1))))
    assert np.array_equal(env.boards.cpu().numpy(), envs["board"][:, :n * n])
    assert_aux(env.aux_fields(), envs, "closed key set")
    st = agent.stats()
    print(f"[closed key set {n}x{n} {path} cache={cache} eps={eps}] {rows} rows, {drops} of {B * k2} updates dropped, "
          f"worst relative Q error {worst:.2e}, longest streak of one action {int(envs['consecutive_count'].max())}")
    assert drops > 0.2 * B * k2                           # most of a 2048 game's states are new
    assert st["drops"] == drops and st["inserts"] == rows == rows1 == agent.table_size()
    assert (path in ("four_call", "mixed") or st["steps"] == B * (k1 + k2)) and agent.check_status() == 0
    if eps == 0.0:        # greedy play on visit rows: an invalid move teaches the next choice -- nobody repeats one action into the stall rule
        assert int(envs["consecutive_count"].max()) <= 60 and st["episodes"] > 0   # (valid repeats of one action happen; 100 invalid ones would end the episode)
    assert pkg._native.claim_timeouts(LIB(pkg)) == 0


@pytest.mark.parametrize("n", [4, 5])
def test_closed_key_set_shared_table_drops_match_oracle(pkg, O, n):
    """One SHARED table, 20 000 envs, epsilon = 1 (the trajectories are the draws'): 40 steps of ordinary learning, then
    the key set is closed for 80 more.  The table's key set afterwards IS the oracle dict's at the moment it froze,
    and the number of dropped updates equals the oracle's count of env-steps taken from a state outside that set --
    a function of the draws alone, whatever the order in which the lanes ran."""
    B, k1, k2, seed, id0 = 20000, 40, 80, 6, 99
    env = pkg.BatchedGame2048Env(B, board_size=n, seed=seed, env_id0=id0, device=DEV)
    agent = pkg.BatchedQLearningAgent(100, exploration_rate=1.0, discount_factor=0.0, capacity_log2=22, seed=seed,
                                      env_id0=id0, device=DEV, board_size=n, freeze_load=None)
    agent.fused_rollout(env, k1)
    agent.frozen = True
    agent.fused_rollout(env, k2)
    envs = O.envs_init(B, n, seed, id0)
    oa = O.Agent(100, 4, 0.1, 0.0, 1.0, n=n)
    O.rollout(envs, oa, k1, seed, id0, 0)
    size1 = len(oa)
    oa.freeze()
    si, _ = O.rollout(envs, oa, k2, seed, id0, k1)
    assert np.array_equal(env.boards.cpu().numpy(), envs["board"][:, :n * n])
    assert_aux(env.aux_fields(), envs, "closed key set, shared")
    st = agent.stats()
    assert len(oa) == size1 == st["inserts"] == agent.table_size()
    assert st["drops"] == si[O.ST_DROPS] == oa.drops > 0.5 * B * k2
    keys, _ = oa.dump()
    _, found = agent.q_values(t8(keys), return_found=True)
    assert bool(found.all()) and agent.check_status() == 0
    print(f"[closed key set {n}x{n}, shared] {size1} rows, {st['drops']} of {B * k2} updates dropped == the oracle's count")


@pytest.mark.parametrize("n,eps", [(4, 0.2), (5, 0.2), (4, 0.0)])
def test_closed_key_set_deterministic_mode_is_bit_exact(pkg, O, n, eps):
    """The deterministic step with Q2048_FLAG_NO_NEW_ROWS on a SHARED table at epsilon 0.2 (and 0: greedy) against the
    oracle's two-phase semantic with a dict that stopped taking keys (float32 rows): boards bit-exact, the WHOLE
    table bit-exact, the same drops.  The envs' visit rows (q2048_det_rollout_cached) cross the call boundary in the
    row cache; every action of a greedy env in a state without a row depends on them."""
    B, k1, k2, seed, id0, lr, gamma = 3000, 30, 90, 41, 10, 0.1, 0.95
    env = pkg.BatchedGame2048Env(B, board_size=n, seed=seed, env_id0=id0, device=DEV)
    agent = pkg.BatchedQLearningAgent(100, learning_rate=lr, discount_factor=gamma, exploration_rate=eps,
                                      capacity_log2=20, seed=seed, env_id0=id0, device=DEV, board_size=n,
                                      freeze_load=None)
    agent.deterministic_rollout(env, k1)
    agent.frozen = True
    agent.deterministic_rollout(env, k2 // 2)
    agent.deterministic_rollout(env, k2 - k2 // 2)
    envs = O.envs_init(B, n, seed, id0)
    oa = O.Agent(100, 4, lr, gamma, eps, n=n, storage_f32=True)
    si1, _ = O.rollout_sync(envs, oa, k1, seed, id0, 0)
    size1 = len(oa)
    oa.freeze()
    si, _ = O.rollout_sync(envs, oa, k2, seed, id0, k1)
    assert np.array_equal(env.boards.cpu().numpy(), envs["board"][:, :n * n])
    assert_aux(env.aux_fields(), envs, "closed key set, deterministic")
    keys, vals = oa.dump()
    got = agent.q_values(t8(keys)).cpu().numpy()
    assert np.array_equal(got, vals.astype(np.float32)), f"{(got != vals.astype(np.float32)).sum()} entries differ"
    st = agent.stats()
    assert len(oa) == size1 == st["inserts"] == agent.table_size()
    assert st["drops"] == si[O.ST_DROPS] == oa.drops > 0 and st["explored"] == si1[O.ST_EXPLORE] + si[O.ST_EXPLORE]
    assert agent.check_status() == 0


def test_line_summaries_of_a_closed_key_set(pkg, O):
    """Q2048_FLAG_LINE_SUMMARY (4x4): q2048_table_summarise writes, into the spare word of every slot, the four 16-bit
    fingerprints of its 128-byte line (0 = empty; bits 48..63 of the key's hash, | 1) -- checked here word by word
    against the raw table -- and the fused rollout decides its lookups from them.  Same results as the slot-by-slot
    probe: with private rows the run with summaries equals the oracle's agents board for board, row for row, drop for
    drop (`line_summaries = False` is the slot-by-slot probe on the same path).  The agent writes them at the first launch after the key set closed and again after anything that may have
    created rows (a launch with the key set open); a 5x5 table never gets them (its second word is a key word)."""
    B, k1, k2, seed, id0, eps, lr, gamma = 192, 60, 120, 5, 900, 0.1, 0.1, 0.95
    m64 = (1 << 64) - 1

    def mix64(x):                                   # q2048::mix64 (csrc/q2048_core.hpp), restated for the check
        x = (x * 0x9E3779B97F4A7C15) & m64
        x ^= x >> 29
        x = (x * 0xBF58476D1CE4E5B9) & m64
        x ^= x >> 32
        return x

    def check_summaries(agent):
        raw = agent.table.cpu().numpy().view(np.uint64).reshape(-1, 4, 4)      # [line, slot, word]
        keys, words = raw[:, :, 0], raw[:, :, 3]
        occupied = np.nonzero(keys.any(axis=1))[0]
        assert (words[keys.any(axis=1) == False] == 0).all()                   # noqa: E712  (an empty line: summary 0)
        for line in occupied[:: max(1, len(occupied) // 400)]:
            want = 0
            for r in range(4):
                k = int(keys[line, r])
                if k:
                    want |= (((mix64(k) >> 48) & 0xFFFF) | 1) << (16 * r)
            assert all(int(w) == want for w in words[line]), (line, want, words[line])
        return len(occupied)

    runs = {}
    for summaries in (True, False):
        env = pkg.BatchedGame2048Env(B, seed=seed, env_id0=id0, device=DEV)
        agent = pkg.BatchedQLearningAgent(100, learning_rate=lr, discount_factor=gamma, exploration_rate=eps,
                                          capacity_log2=15, seed=seed, env_id0=id0, device=DEV, freeze_load=None)
        agent.line_summaries = summaries
        agent.fused_rollout(env, k1)
        assert not agent._summarised and not agent.table.cpu().numpy().view(np.uint64).reshape(-1, 4)[:, 3].any()
        agent.frozen = True
        agent.fused_rollout(env, k2 // 2)
        assert agent._summarised == summaries
        if summaries:
            assert check_summaries(agent) > 1000
            # the key set opens again (by hand): rows are created, the summaries are stale -- and are written again
            # before the next launch that uses them
            agent.frozen = False
            agent.fused_rollout(env, 4)
            assert not agent._summarised
            agent.frozen = True
            agent.fused_rollout(env, k2 - k2 // 2 - 4)
            assert agent._summarised
            check_summaries(agent)
        else:
            agent.frozen = False
            agent.fused_rollout(env, 4)
            agent.frozen = True
            agent.fused_rollout(env, k2 - k2 // 2 - 4)
        runs[summaries] = agent.stats()
        assert agent.check_status() == 0 and runs[summaries]["steps"] == B * (k1 + k2)
    assert runs[True]["drops"] > 0 and runs[False]["drops"] > 0
    # private rows: the summaries' run against the oracle, value for value
    env = pkg.BatchedGame2048Env(B, seed=seed, env_id0=id0, device=DEV)
    agent = pkg.BatchedQLearningAgent(100, learning_rate=lr, discount_factor=gamma, exploration_rate=eps,
                                      capacity_log2=17, seed=seed, env_id0=id0, device=DEV, independent=True,
                                      freeze_load=None)
    agent.fused_rollout(env, k1)
    agent.frozen = True
    agent.fused_rollout(env, k2 // 3)
    agent.fused_rollout(env, k2 - k2 // 3)
    assert agent._summarised
    envs = O.envs_init(B, 4, seed, id0)
    drops = 0
    for i in range(B):
        oa = O.Agent(100, 4, lr, gamma, eps)
        O.rollout(envs[i:i + 1], oa, k1, seed, id0 + i, 0)
        oa.freeze()
        O.rollout(envs[i:i + 1], oa, k2, seed, id0 + i, k1)
        drops += oa.drops
        keys, vals = oa.dump()
        got, found = agent.q_values(t8(keys), env_id=id0 + i, return_found=True)
        assert bool(found.all()) and np.allclose(got.cpu().numpy(), vals, rtol=1e-5, atol=1e-6), i
    assert np.array_equal(env.boards.cpu().numpy(), envs["board"][:, :16]) and agent.stats()["drops"] == drops > 0
    # 5x5: the flag is ignored, the second word stays a key word
    env5 = pkg.BatchedGame2048Env(64, board_size=5, seed=seed, device=DEV)
    a5 = pkg.BatchedQLearningAgent(100, exploration_rate=0.3, capacity_log2=14, seed=seed, device=DEV, board_size=5,
                                   freeze_load=None)
    a5.fused_rollout(env5, 40)
    before = a5.table.clone()
    a5.frozen = True
    a5.experiment_bits = pkg._native.FLAG_LINE_SUMMARY          # (passed by hand: the agent itself never does on 5x5)
    a5.fused_rollout(env5, 40)
    assert not a5._summarised and a5.check_status() == 0
    w0, w1 = (t.cpu().numpy().view(np.uint64).reshape(-1, 4) for t in (before, a5.table))
    assert np.array_equal(w0[:, 0], w1[:, 0]) and np.array_equal(w0[:, 3], w1[:, 3])     # keys and second words untouched


def test_growing_table_freezes_at_its_largest_capacity(pkg):
    """capacity_log2="auto" with max_capacity_log2 = 2^20: the table grows 2^16 -> 2^18 -> 2^20 like the defaultdict
    (every growth checks rows moved == rows created), then -- it cannot grow any more -- closes its key set at
    freeze_load; the run goes on at the young table's speed class, `verify_table` still holds, a checkpoint of the
    frozen learner loads into a fresh agent that freezes again at its first launch."""
    import warnings

    B, S, seed = 8192, 8, 3
    env = pkg.BatchedGame2048Env(B, seed=seed, device=DEV)
    agent = pkg.BatchedQLearningAgent(100, exploration_rate=0.9, capacity_log2="auto", initial_capacity_log2=16,
                                      max_capacity_log2=20, seed=seed, device=DEV, freeze_load=0.5)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        for _ in range(40):
            agent.fused_rollout(env, S)
    agent.finish_growth()
    assert agent.capacity_log2 == 20 and [g["to_log2"] for g in agent.growths][-1] == 20
    assert agent.frozen and sum("takes no new rows" in str(w.message) for w in caught) == 1
    check = agent.verify_table()
    assert 0.5 <= check["load"] <= 0.5 + 2.0 * B * S / (1 << 20) and check["rows"] == agent.frozen_at["rows"]
    st = agent.stats()
    assert st["drops"] > 0 and st["inserts"] == check["rows"] and agent.check_status() == 0
    sd = agent.state_dict()
    fresh = pkg.BatchedQLearningAgent(100, exploration_rate=0.9, capacity_log2=20, seed=seed, device=DEV, freeze_load=0.5)
    fresh.load_state_dict(sd)
    assert not fresh.frozen
    env.ctr = fresh.ctr
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fresh.fused_rollout(env, S)
    assert fresh.frozen and fresh.table_size() == check["rows"]


@pytest.mark.parametrize("n,closed", [(4, False), (5, False), (4, True), (5, True)])
def test_checkpoint_resume_is_bit_exact(pkg, n, closed):
    """state_dict / load_state_dict: stop after 60 steps, reload into fresh
    objects with a DIFFERENT table capacity, continue 40 steps == the uninterrupted run.
    `closed`: the key set closes (Q2048_FLAG_NO_NEW_ROWS) 30 steps before the checkpoint, at epsilon 0.05 -- the
    envs' VISIT ROWS are part of the run: they travel in the checkpoint (`visit_rows`: the row cache's bytes,
    re-bound to the restored table by q2048_rowcache_rebind), and a resume without them plays other moves."""
    B, seed, id0, eps = 300, 9, 4242, (0.05 if closed else 0.3)

    def mk(cap):
        e = pkg.BatchedGame2048Env(B, board_size=n, seed=seed, env_id0=id0, device=DEV)
        a = pkg.BatchedQLearningAgent(100, exploration_rate=eps, discount_factor=0.95, capacity_log2=cap,
                                      seed=seed, env_id0=id0, device=DEV, independent=True, board_size=n,
                                      freeze_load=None)
        return e, a

    e1, a1 = mk(17)
    if closed:
        a1.fused_rollout(e1, 30)
        a1.frozen = True
        a1.fused_rollout(e1, 30)
    else:
        a1.fused_rollout(e1, 60)
    sd_env, sd_agent = e1.state_dict(), a1.state_dict()
    assert ("visit_rows" in sd_agent) == closed
    a1.decay_exploration(0); eps_after = a1.epsilon
    a1.fused_rollout(e1, 40)

    def resume(sd):
        e2, a2 = mk(19)
        e2.load_state_dict(sd_env); a2.load_state_dict(sd)
        assert a2.table_size() == len(sd["q"]) and (e2.ctr, a2.ctr) == (60, 60)
        a2.frozen = closed
        a2.decay_exploration(0); assert a2.epsilon == eps_after
        a2.fused_rollout(e2, 40)
        return e2, a2

    e2, a2 = resume(sd_agent)
    assert torch.equal(e1.boards, e2.boards) and torch.equal(e1.aux, e2.aux)
    k1, q1 = a1.export_rows(); k2, q2 = a2.export_rows()
    k1 = k1.reshape(len(q1), -1); k2 = k2.reshape(len(q2), -1)
    o1 = np.lexsort(k1.T[::-1]); o2 = np.lexsort(k2.T[::-1])
    assert np.array_equal(k1[o1], k2[o2]) and np.array_equal(q1[o1], q2[o2])
    s1, s2 = a1.stats(), a2.stats()
    for k in ("steps", "episodes", "valid_moves", "score_sum", "explored", "drops", "inserts"):
        assert s1[k] == s2[k], k
    if closed:
        assert s1["drops"] > 0 and s1["inserts"] == len(q1) == len(sd_agent["q"])
        e3, _ = resume({k: v for k, v in sd_agent.items() if k != "visit_rows"})
        assert not torch.equal(e1.boards, e3.boards)      # (the test has teeth: the visit rows decide moves)


def test_export_dict_is_the_reference_table(pkg, O):
    """A GPU-trained table in the reference's own form {tuple-of-tuples: float64[4]}."""
    steps, seed, id0 = 800, 12, 99
    env = pkg.BatchedGame2048Env(1, seed=seed, env_id0=id0, device=DEV)
    agent = pkg.BatchedQLearningAgent(100, exploration_rate=0.2, discount_factor=0.99,
                                      capacity_log2=14, seed=seed, env_id0=id0, device=DEV)
    agent.fused_rollout(env, steps)
    envs = O.envs_init(1, 4, seed, id0)
    oa = O.Agent(100, 4, 0.1, 0.99, 0.2)
    O.rollout(envs, oa, steps, seed, id0, 0)
    d = agent.export_dict()
    keys, vals = oa.dump()
    assert len(d) == len(keys)
    for k, v in zip(keys, vals):
        state = tuple(tuple(int(x) for x in r) for r in pkg.boards_to_raw(k))
        assert state in d and d[state].dtype == np.float64
        assert np.allclose(d[state], v, rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------------------------------------
# BASELINE configs[1]: 65,536 boards, flat-array Q over row-tuple features.  Not the reference's
# learner: checked against the oracle's own restatement of it.
# ---------------------------------------------------------------------------------------------
def test_row_tuple_single_env_matches_oracle(pkg, O):
    """One env = the sequential learner: same actions, boards, weights (fused and 4-call API)."""
    steps, seed, id0, eps, lr, gamma = 3000, 14, 555, 0.15, 0.1, 0.99
    envs = O.envs_init(1, 4, seed, id0)
    oa = O.RowTupleAgent(lr, gamma, eps)
    si, sf = oa.rollout(envs, steps, seed, id0, 0)
    for mode in ("fused", "unfused"):
        env = pkg.BatchedGame2048Env(1, seed=seed, env_id0=id0, device=DEV)
        agent = pkg.BatchedRowTupleAgent(100, learning_rate=lr, discount_factor=gamma,
                                         exploration_rate=eps, seed=seed, env_id0=id0, device=DEV)
        if mode == "fused":
            for k in (1, 999, 2000):
                agent.fused_rollout(env, k)
        else:
            _unfused_loop(pkg, env, agent, steps)
        assert np.array_equal(env.boards.cpu().numpy(), envs["board"][:, :16]), mode
        assert_aux(env.aux_fields(), envs, "row tuple " + mode)
        w, ow = agent.weights.cpu().numpy(), oa.weights()
        assert np.array_equal(w != 0, ow != 0), mode                 # same entries touched
        assert np.allclose(w, ow, rtol=1e-5, atol=1e-6), mode
        if mode == "fused":
            st = agent.stats()
            assert st["steps"] == steps and st["episodes"] == si[O.ST_EPISODES]
            assert st["valid_moves"] == si[O.ST_VALID] and st["explored"] == si[O.ST_EXPLORE]
        assert agent.check_status() == 0


def test_row_tuple_65536_boards_properties(pkg, O):
    """configs[1] size.  eps = 1: trajectories are exact whatever the racing weight writes do;
    the weights stay bounded and track the sequential oracle's (same entries touched)."""
    B, steps, seed = 65536, 64, 31
    env = pkg.BatchedGame2048Env(B, seed=seed, device=DEV)
    agent = pkg.BatchedRowTupleAgent(100, exploration_rate=1.0, learning_rate=0.05,
                                     discount_factor=0.0, seed=seed, device=DEV)
    agent.fused_rollout(env, steps)
    envs = O.envs_init(B, 4, seed, 0)
    oa = O.RowTupleAgent(0.05, 0.0, 1.0)
    si, sf = oa.rollout(envs, steps, seed, 0, 0)
    assert np.array_equal(env.boards.cpu().numpy(), envs["board"][:, :16])
    st = agent.stats()
    assert st["steps"] == B * steps and st["episodes"] == si[O.ST_EPISODES]
    assert st["valid_moves"] == si[O.ST_VALID] and st["score_sum"] == si[O.ST_SCORE]
    w, ow = agent.weights.cpu().numpy(), oa.weights()
    assert np.array_equal(w != 0, ow != 0)              # exactly the same weights were touched
    # every write is one valid TD step from a value some lane read (last writer wins), so the
    # weights stay within the reward scale; the sequential oracle applies far more updates to the
    # hot entries, so values are compared loosely
    assert np.isfinite(w).all() and np.abs(w).max() <= 10.0
    assert np.corrcoef(w.ravel(), ow.ravel())[0, 1] > 0.5
    q = agent.q_values(env.boards[:1000]).cpu().numpy()
    assert np.isfinite(q).all() and agent.check_status() == 0


def test_episode_log_matches_reference_csv_rows(pkg):
    """The device episode log against the reference's own transcript (golden G6: 30 episodes of
    Agent/main.py:80-109 with the draws injected): one record per finished episode carrying the
    columns of debug_log.csv (main.py:62) -- Episode, Action, Q-Values, Reward, Total-Reward,
    Max Value -- plus the epsilon schedule applied per episode (:109)."""
    g = load_npz("g6_episodes_seed0.npz")
    seed, id0, E = int(g["seed"]), int(g["env_id0"]), int(g["E"])
    env = pkg.BatchedGame2048Env(1, seed=seed, env_id0=id0, device=DEV)
    agent = pkg.BatchedQLearningAgent(E, learning_rate=float(g["lr"]), discount_factor=float(g["gamma"]),
                                      exploration_rate=float(g["eps0"]), capacity_log2=16, seed=seed,
                                      env_id0=id0, device=DEV)
    log = pkg.EpisodeLog(64, device=DEV)
    finished, recs, eps_trace, last_rows = 0, [], [], []
    for t in range(int(g["steps"])):
        prev = env.boards.clone()
        agent.fused_rollout(env, 1, episode_log=log)
        if int(log.count.item()) > 0:
            r = log.drain()
            assert len(r) == 1
            recs.append(r[0])
            last_rows.append(agent.q_values(prev).cpu().numpy()[0])
            agent.decay_exploration(finished)                           # Agent/main.py:109
            eps_trace.append(agent.epsilon)
            finished += 1
    ends = np.flatnonzero(g["dones"])
    assert finished == len(ends) == len(g["ep_returns"])
    assert np.array_equal(np.array(eps_trace), g["eps_trace"])          # bit-identical schedule
    for e, (rec, idx) in enumerate(zip(recs, ends)):
        row = pkg.EpisodeLog.csv_row(rec)
        assert row[0] == e and rec["env_id"] == id0
        assert row[1] == g["actions"][idx]                              # Action
        assert np.float32(row[3]) == np.float32(g["rewards"][idx])      # Reward
        assert np.isclose(row[4], g["ep_returns"][e], rtol=1e-5)        # Total-Reward
        assert row[5] == g["maxes"][idx]                                # Max Value
        assert rec["score"] == g["ep_scores"][e]
        assert np.array_equal(np.asarray(rec["q"]), last_rows[e])       # Q-Values = live row of `state`
    assert np.array_equal(env.boards.cpu().numpy(), g["final_boards"])
    # the run summary in the layout of plots/summary_statistics_cleaned.csv, from the device
    # records, against the same aggregate of the reference's transcript
    got_row = pkg.summarize_records("g6", np.array(recs))
    want_row = pkg.summarize_episodes("g6", g["actions"][ends], g["rewards"][ends], g["maxes"][ends])
    assert got_row[0] == "g6" and got_row[3:] == want_row[3:]           # Max_Value, Action_0..3
    assert np.allclose(got_row[1:3], want_row[1:3], rtol=1e-6)          # Avg / Std of float32 rewards
    # the whole Q-table of the run against the reference's dict
    got = agent.q_values(t8(g["q_keys"])).cpu().numpy()
    assert np.allclose(got, g["q_vals"], rtol=1e-5, atol=1e-6) and agent.table_size() == len(g["q_keys"])


def test_episode_log_batched_overflow_and_order(pkg):
    B, steps = 2048, 400
    env = pkg.BatchedGame2048Env(B, seed=3, device=DEV)
    agent = pkg.BatchedQLearningAgent(10, exploration_rate=1.0, capacity_log2=22, seed=3, device=DEV)
    log = pkg.EpisodeLog(1000, device=DEV)
    agent.fused_rollout(env, steps, episode_log=log)
    st = agent.stats()
    rec = log.drain()
    assert st["episodes"] > 1000 and len(rec) == 1000 and log.lost == st["episodes"] - 1000
    assert rec["env_id"].max() < B and np.all(rec["total_return"] != 0)
    # per env, records arrive in episode order
    for e in np.unique(rec["env_id"])[:50]:
        ep = rec["episode"][rec["env_id"] == e]
        assert np.all(np.diff(ep.astype(np.int64)) > 0)


# ---------------------------------------------------------------------------------------------
# bridges to the reference's DQN front-end
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [4, 5])
def test_legal_moves_mask(pkg, O, n):
    """bit a == "move a changes the board", i.e. the reference's trial-move loop
    (mainDQL_CNN_step2.py:168-174), against the oracle's move()."""
    B = 3000
    env = pkg.BatchedGame2048Env(B, board_size=n, seed=8, device=DEV)
    acts = torch.randint(0, 4, (B,), dtype=torch.uint8, device=DEV)
    for _ in range(60):                       # mid-game boards, some of them stuck in a direction
        env.step(acts)
    before = env.boards.clone()
    mask = env.legal_moves().cpu().numpy()
    assert torch.equal(before, env.boards)    # a probe, nothing is modified
    boards = before.cpu().numpy()
    want = np.array([sum(int(O.move(b, a, n=n)[2]) << a for a in range(4)) for b in boards], np.uint8)
    assert np.array_equal(mask, want) and len(set(want.tolist())) > 4
    g = load_npz("g3_game_over.npz")          # dead boards of the reference: no legal move
    if n == 4:
        e2 = pkg.BatchedGame2048Env(len(g["boards"]), seed=0, device=DEV)
        e2.boards.copy_(t8(g["boards"]))
        m2 = e2.legal_moves().cpu().numpy()
        full = (g["boards"] != 0).all(axis=1)
        assert np.array_equal((m2 == 0) & full, g["over"].astype(bool))


def test_encode_onehot_matches_reference_encoder(pkg):
    """Dqn8TestNOPERCNN.py:271-277 restated in numpy: one_hot(log2 tile, depth 16) as [16,4,4]."""
    B = 1000
    env = pkg.BatchedGame2048Env(B, seed=2, device=DEV)
    acts = torch.randint(0, 4, (B,), dtype=torch.uint8, device=DEV)
    for _ in range(80):
        env.step(acts)
    env.boards[0, 5] = 16                     # a 2^16 tile: tf.one_hot(16, depth=16) is all zeros
    boards = env.boards.cpu().numpy()
    want = (boards[:, None, :] == np.arange(16, dtype=np.uint8)[None, :, None]).reshape(B, 16, 4, 4)
    f32 = env.encode_onehot().cpu().numpy()
    assert f32.dtype == np.float32 and np.array_equal(f32, want.astype(np.float32))
    assert f32[0, :, 1, 1].sum() == 0 and np.all(f32[1:].sum(axis=1) == 1)
    bf = env.encode_onehot(torch.bfloat16)
    assert bf.dtype == torch.bfloat16 and np.array_equal(bf.float().cpu().numpy(), want.astype(np.float32))


def test_tile_overflow_is_reported(pkg):
    """A tile above 2^15 does not fit the 64-bit state key: the status word says so (the env
    itself keeps playing exactly; only the agent's key aliases)."""
    env = pkg.BatchedGame2048Env(64, seed=1, device=DEV)
    agent = pkg.BatchedQLearningAgent(10, exploration_rate=1.0, capacity_log2=12, seed=1, device=DEV)
    board = np.zeros(16, dtype=np.uint8)
    board[0] = board[1] = 15                       # two 32768 tiles: moving left makes 65536
    env.boards[7].copy_(torch.from_numpy(board).to(DEV))
    acts = torch.zeros(64, dtype=torch.uint8, device=DEV)
    _, r, d, mx = env.step(acts)
    assert int(env.boards[7, 0]) == 16 and int(mx[7]) == 65536 and int(env.score[7]) == 65536
    assert agent.check_status() == 0
    agent.q_values(env.boards)
    assert agent.check_status() & pkg._native.STATUS_TILE_OVERFLOW


def test_steps_counter_and_argument_checks(pkg):
    env = pkg.BatchedGame2048Env(8, seed=1, device=DEV)
    agent = pkg.BatchedQLearningAgent(10, capacity_log2=10, seed=1, device=DEV)
    agent.fused_rollout(env, 3)
    assert (env.ctr, agent.ctr) == (3, 3)
    env.step(torch.zeros(8, dtype=torch.uint8, device=DEV))
    with pytest.raises(ValueError, match="out of step"):
        agent.fused_rollout(env, 1)
    other = pkg.BatchedQLearningAgent(10, capacity_log2=10, seed=2, device=DEV)
    with pytest.raises(ValueError, match="share seed"):
        other.fused_rollout(pkg.BatchedGame2048Env(8, seed=1, device=DEV), 1)
    with pytest.raises(ValueError):
        env.step(torch.zeros(7, dtype=torch.uint8, device=DEV))
    with pytest.raises(ValueError):
        agent.q_values(torch.zeros((4, 25), dtype=torch.uint8, device=DEV))
    q0, f0 = agent.q_values(torch.zeros((0, 16), dtype=torch.uint8, device=DEV), return_found=True)   # an empty batch is not an error
    assert tuple(q0.shape) == (0, 4) and tuple(f0.shape) == (0,)
    none = torch.zeros((0, 16), dtype=torch.uint8, device=DEV)
    ctr = agent.ctr
    assert tuple(agent.choose_action(none).shape) == (0,) and agent.ctr == ctr
    agent.update_q_value(none, none[:, 0], torch.zeros(0, device=DEV), none, none[:, 0])
    with pytest.raises(pkg.NativeError):
        pkg._native.check(LIB(pkg).q2048_fused_rollout(
            env.boards.data_ptr(), env.aux.data_ptr(), agent.table.data_ptr(), agent.capacity_log2, 8, 4,
            -1, 0.5, 0.1, 0.9, 0, 0, 0, 0, None, None, agent.status.data_ptr(), None), "neg steps")


@pytest.mark.parametrize("strict,freeze_after", [(False, None), (True, None), (False, 20), (True, 20)])
def test_shared_table_writes_are_legitimate_values(pkg, O, strict, freeze_after):
    """Shared table, lanes racing on common states.  With lr = 1, gamma = 0 the update writes
    Q[s][a] = reward, so whatever the interleaving every stored value must be EXACTLY one of the
    float32 rewards the oracle saw for that (state, action); untouched entries stay 0; no row may
    exist for a state the oracle never visited.  (Both write modes: store and compare-and-swap.)
    `freeze_after`: the key set closes after that many steps (Q2048_FLAG_NO_NEW_ROWS) -- the racing lanes go on
    writing legitimate values into the rows that exist, and the table holds exactly the oracle's closed key set."""
    B, steps, seed, id0 = 20000, 60, 33, 400
    env = pkg.BatchedGame2048Env(B, seed=seed, env_id0=id0, device=DEV)
    agent = pkg.BatchedQLearningAgent(100, learning_rate=1.0, discount_factor=0.0, exploration_rate=1.0,
                                      capacity_log2=22, seed=seed, env_id0=id0, device=DEV, strict_td=strict,
                                      freeze_load=None)
    if freeze_after is None:
        agent.fused_rollout(env, steps)
    else:
        agent.fused_rollout(env, freeze_after)
        agent.frozen = True
        agent.fused_rollout(env, steps - freeze_after)
    envs = O.envs_init(B, 4, seed, id0)
    oa = O.Agent(100, 4, 1.0, 0.0, 1.0)
    shifts = (4 * np.arange(16, dtype=np.uint64))
    seen = {}
    for t in range(steps):
        if t == freeze_after:
            oa.freeze()
        keys = (envs["board"][:, :16].astype(np.uint64) << shifts).sum(axis=1)
        si, sf, a, r, d = O.rollout(envs, oa, 1, seed, id0, t, record=True)
        r32 = r[0].astype(np.float32)
        for k, act, rew in zip(keys.tolist(), a[0].tolist(), r32.tolist()):
            seen.setdefault((k, act), set()).add(rew)
    assert np.array_equal(env.boards.cpu().numpy(), envs["board"][:, :16])
    dk, dq = agent.export_rows()
    visited = {k for k, _ in seen} | set(((envs["board"][:, :16].astype(np.uint64) << shifts).sum(axis=1)).tolist())
    shared = sum(1 for v in seen.values() if len(v) > 1)
    assert shared > 1000                       # the race is real: many (s, a) saw several rewards
    bad = 0
    for k, row in zip(dk.tolist(), dq):
        assert k in visited
        for act in range(4):
            vals = seen.get((k, act))
            if vals is None:
                bad += row[act] != 0.0
            else:
                bad += float(row[act]) not in vals
    assert bad == 0 and len(dk) == len(oa)
    assert agent.stats()["drops"] == oa.drops and (oa.drops > 0) == (freeze_after is not None) and agent.check_status() == 0


def test_c_host_program_drives_the_abi(pkg, tmp_path):
    """The boundary is a plain C ABI: examples/rollout_host.c (hipMalloc + q2048_* calls, no
    Python, no torch) must reproduce the Python host's run."""
    import shutil
    import subprocess
    from conftest import REPO
    gcc, rocm = shutil.which("gcc"), os.environ.get("ROCM_PATH", "/opt/rocm")
    if gcc is None or not os.path.exists(os.path.join(rocm, "include", "hip", "hip_runtime_api.h")):
        pytest.skip("gcc or the HIP runtime headers are not available")
    exe = str(tmp_path / "rollout_host")
    libdir = os.path.dirname(pkg._native.LIB_PATH)
    subprocess.run([gcc, "-std=c11", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(rocm, "include"),
                    "-I", os.path.join(REPO, "include"), os.path.join(REPO, "examples", "rollout_host.c"),
                    "-o", exe, "-L", libdir, "-lq2048_hip", "-L", os.path.join(rocm, "lib"), "-lamdhip64",
                    f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{os.path.join(rocm, 'lib')}"], check=True)
    B, steps, seed, cap, eps = 5000, 70, 17, 21, 1.0
    env = pkg.BatchedGame2048Env(B, seed=seed, device=DEV)
    agent = pkg.BatchedQLearningAgent(100, learning_rate=0.1, discount_factor=0.99, exploration_rate=eps,
                                      capacity_log2=cap, seed=seed, device=DEV)
    agent.fused_rollout(env, steps)
    st = agent.stats()
    # one launch through q2048_fused_rollout; then launches of 16 steps through q2048_fused_rollout_opts
    # with a row cache and the statistics read from the host-side mirror
    # ... and the same launches on a table that grows fourfold before launch 2, off the critical path
    # (q2048_table_reserve / _grow_begin / _grow_commit / _grow_finish from plain C): the same run
    # ... and on a table that CLOSES ITS KEY SET before launch 2 (Q2048_FLAG_NO_NEW_ROWS; q2048_table_summarise +
    # Q2048_FLAG_LINE_SUMMARY from plain C): the Python host's run with `agent.frozen = True` from that launch on
    env_c = pkg.BatchedGame2048Env(B, seed=seed, device=DEV)
    agent_c = pkg.BatchedQLearningAgent(100, learning_rate=0.1, discount_factor=0.99, exploration_rate=eps,
                                        capacity_log2=cap, seed=seed, device=DEV, freeze_load=None)
    for launch, at in enumerate(range(0, steps, 16)):
        agent_c.frozen = launch >= 2
        agent_c.fused_rollout(env_c, min(16, steps - at))
    st_c = agent_c.stats()
    assert st_c["drops"] > 0 and agent_c._summarised
    for per_launch, grow_at, close_at in ((0, None, None), (16, None, None), (16, 2, None), (16, -1, 2)):
        argv = [exe, str(B), str(steps), str(seed), str(cap), str(eps), str(per_launch)] + (
            [str(grow_at)] if grow_at is not None else []) + ([str(close_at)] if close_at is not None else [])
        out = subprocess.run(argv, check=True, capture_output=True, text=True).stdout
        got = json.loads(out.strip().splitlines()[-1])
        want, want_agent, want_env = (st, agent, env) if close_at is None else (st_c, agent_c, env_c)
        assert got["capacity_log2"] == cap + (2 if grow_at is not None and grow_at >= 0 else 0)
        for k in ("steps", "episodes", "valid_moves", "score_sum", "inserts", "drops", "explored"):
            assert got[k] == want[k], (per_launch, grow_at, close_at, k)
        assert got["rows"] == want_agent.table_size() and got["status"] == 0
        assert got["board0"] == want_env.boards[0].cpu().tolist()
        assert np.isclose(got["return_sum"], want["return_sum"], rtol=1e-9)


@pytest.mark.parametrize("n,B,steps,sort_bits", [(4, 3000, 80, 0), (5, 1500, 60, 0), (4, 60000, 24, 0),
                                                  (4, 3000, 80, 3), (5, 1500, 60, 8), (4, 60000, 24, 63),
                                                  (4, 60000, 24, 10), (4, 2500, 330, 0), (5, 2100, 300, 0),
                                                  (4, 60000, 24, 64), (4, 5000, 40, 64 + 8)])
def test_deterministic_mode_matches_oracle_on_a_shared_table(pkg, O, monkeypatch, n, B, steps, sort_bits):
    """Shared table, lanes meeting on common states, epsilon < 1 (actions depend on Q): the
    deterministic mode equals the oracle's two-phase semantic with float32 rows (the oracle's
    `storage_f32` option, pinned against G6/G7) -- boards bit-exact and the WHOLE Q-table BIT-EXACT:
    both sides fold a group's updates in env order with the same double operations (no fused
    multiply-add) and round to float32 once per group and step -- and two runs give bit-identical
    tables.  `sort_bits` != 0 runs on the measurement build (tools/variants/libq2048_hip_exp.so, the
    same sources with -DQ2048_EXPERIMENTS): the updates are then sorted by the low 3 / 8 / 10 bits
    of the 16-bit (state, action) hash, which makes every run of the sorted array a crowd of
    different groups (each told apart by the full word) or mixes long runs with crowded ones, or by
    the whole word (63: a run is a group); + 64 selects the partition with its separate scan launch,
    the path batches beyond 8 Mi updates take (the default builds the prefix from group sums)."""
    seed, id0, eps, lr, gamma = 41, 10, 0.2, 0.1, 0.95
    cells = n * n
    if sort_bits:
        monkeypatch.setattr(pkg._native, "_lib", pkg._native.load(pkg._native.build_experiments()))

    def run():
        env = pkg.BatchedGame2048Env(B, board_size=n, seed=seed, env_id0=id0, device=DEV)
        agent = pkg.BatchedQLearningAgent(100, learning_rate=lr, discount_factor=gamma, exploration_rate=eps,
                                          capacity_log2=22, seed=seed, env_id0=id0, device=DEV, board_size=n)
        agent.experiment_bits = sort_bits << 8
        agent.deterministic_rollout(env, steps // 2)          # two calls: the counter carries over
        agent.deterministic_rollout(env, steps - steps // 2)
        return env, agent

    env, agent = run()
    envs = O.envs_init(B, n, seed, id0)
    if B >= 60000:      # many envs on the few opening states: groups longer than one lane walks alone
        _, first = np.unique(envs["board"][:, :cells], axis=0, return_counts=True)
        assert first.max() > 200
    oa = O.Agent(100, 4, lr, gamma, eps, n=n, storage_f32=True)
    si, sf = O.rollout_sync(envs, oa, steps, seed, id0, 0)
    assert np.array_equal(env.boards.cpu().numpy(), envs["board"][:, :cells])
    assert_aux(env.aux_fields(), envs, "deterministic")
    keys, vals = oa.dump()
    got = agent.q_values(t8(keys)).cpu().numpy()
    assert np.allclose(got, vals, rtol=1e-5, atol=1e-6)
    assert np.array_equal(got, vals.astype(np.float32)), f"{(got != vals.astype(np.float32)).sum()} entries differ in the last bit"
    shared = int((np.abs(vals) > 0).sum(axis=1).max())
    st = agent.stats()
    assert st["steps"] == B * steps and st["episodes"] == si[O.ST_EPISODES] and st["explored"] == si[O.ST_EXPLORE]
    if steps >= 300:    # long enough for episodes to end: a new episode's first state is probed, not carried
        assert st["episodes"] > 0
    assert st["inserts"] == agent.table_size() == len(oa) and st["drops"] == 0 and shared >= 2
    assert len(oa) < 0.9 * B * steps          # lanes really do share states
    env2, agent2 = run()
    assert torch.equal(env.boards, env2.boards)
    k1, q1 = agent.export_rows(); k2, q2 = agent2.export_rows()
    k1 = k1.reshape(len(q1), -1); k2 = k2.reshape(len(q2), -1)
    o1 = np.lexsort(k1.T[::-1]); o2 = np.lexsort(k2.T[::-1])
    assert np.array_equal(k1[o1], k2[o2]) and np.array_equal(q1[o1], q2[o2])      # bit-identical
    assert agent.check_status() == 0


@pytest.mark.parametrize("eps", [1.0, 0.95, 0.2])
def test_deterministic_mode_1m_boards_matches_oracle(pkg, O, eps):
    """The bench's own size on ONE shared table against the sequential semantic
    (Agent/main.py:34-43 fed in env order): 1 048 576 boards, 2^30 slots, 8 deterministic steps from
    reset (thousands of lanes per opening state: long runs, one wave each), 256 steps of random play
    with no learner (the bench's input synthesis), 8 more steps on mid-game boards (about 16 distinct
    groups per sorted run).  512 sort tiles: `k_sort_scan` walks its rows in two chunks.
      eps = 1     against the unmodified float64 oracle: actions come from draws alone, so the
                  trajectories are exact whatever the table holds; whole table within 1e-5.
      eps < 1     actions depend on argmax over rows that thousands of lanes share.  The float64
                  oracle breaks mirror-state ties differently from a float32 table (values 2e-14
                  apart in float64 are equal in float32), so the checker is the oracle with float32
                  rows (`storage_f32`, pinned against G6/G7): boards bit-exact, table BIT-EXACT.
    inserts == len(q_table) == the oracle's dict size in every case."""
    B, seed, id0, lr, gamma, cap = 1 << 20, 17, 3, 0.1, 0.99, 30
    release_cached_device_memory()
    env = pkg.BatchedGame2048Env(B, seed=seed, env_id0=id0, device=DEV)
    agent = pkg.BatchedQLearningAgent(100, learning_rate=lr, discount_factor=gamma, exploration_rate=eps,
                                      capacity_log2=cap, seed=seed, env_id0=id0, device=DEV, placement="plain")
    envs = O.envs_init(B, 4, seed, id0)
    oa = O.Agent(100, 4, lr, gamma, eps, storage_f32=eps < 1.0)
    oa.reserve(14 << 20)
    threads = max(1, min(64, (os.cpu_count() or 8)))

    def check(what):
        assert np.array_equal(env.boards.cpu().numpy(), envs["board"][:, :16]), what
        assert_aux(env.aux_fields(), envs, what)
        keys, vals = oa.dump()
        got = agent.q_values(t8(keys)).cpu().numpy()
        assert np.allclose(got, vals, rtol=1e-5, atol=1e-6), what
        if eps < 1.0:
            ne = int((got != vals.astype(np.float32)).sum())
            assert ne == 0, f"{what}: {ne} of {got.size} Q entries differ in the last bit"
        st = agent.stats()
        assert st["inserts"] == agent.table_size() == len(oa) and st["drops"] == 0, what
        return st, vals

    agent.deterministic_rollout(env, 8)
    si, _ = O.rollout_sync(envs, oa, 8, seed, id0, 0)
    st, vals = check("from reset")
    _, first = np.unique(O.envs_init(B, 4, seed, id0)["board"][:, :16], axis=0, return_counts=True)
    assert first.max() > 64 * 16                      # opening states shared by thousands of lanes
    assert st["steps"] == 8 * B and st["explored"] == si[O.ST_EXPLORE]
    # input synthesis as bench.py does it: random play, no learner, the table untouched
    agent.epsilon = 1.0
    agent.fused_rollout(env, 256, play_only=True)
    agent.epsilon = eps
    O.rollout_mt(envs, None, 256, seed, id0, 8, threads=threads)
    rows_before = len(oa)
    assert agent.table_size() == rows_before
    agent.deterministic_rollout(env, 8)
    si2, _ = O.rollout_sync(envs, oa, 8, seed, id0, 264)
    st, vals = check("mid-game")
    assert st["episodes"] > 1000 and len(oa) - rows_before > 4 * B     # mid-game: mostly new states
    assert agent.check_status() == 0
    if eps == 0.2:
        # ... and on with the KEY SET CLOSED (Q2048_FLAG_NO_NEW_ROWS, q2048_det_rollout_cached): most mid-game states
        # have no row, four lanes of five are greedy -- their visit rows decide the actions, and cross a call boundary
        rows_closed = len(oa)
        agent.frozen = True
        oa.freeze()
        agent.deterministic_rollout(env, 3)
        agent.deterministic_rollout(env, 3)
        si3, _ = O.rollout_sync(envs, oa, 6, seed, id0, 272)
        assert np.array_equal(env.boards.cpu().numpy(), envs["board"][:, :16]), "closed key set"
        assert_aux(env.aux_fields(), envs, "closed key set")
        keys, vals = oa.dump()
        got = agent.q_values(t8(keys)).cpu().numpy()
        ne = int((got != vals.astype(np.float32)).sum())
        assert ne == 0, f"closed key set: {ne} of {got.size} Q entries differ in the last bit"
        st3 = agent.stats()
        assert st3["inserts"] == agent.table_size() == len(oa) == rows_closed
        assert st3["drops"] == si3[O.ST_DROPS] == oa.drops > 2 * B and agent.check_status() == 0
    del agent, env
    release_cached_device_memory()


@pytest.mark.parametrize("B", [1, 63, 2047, 2048, 2049, 4097, 6144])
def test_deterministic_mode_batch_edges(pkg, O, B):
    """The partition works on tiles of 2048 updates: batches of one update, of less than a wave, of
    exactly one and exactly three tiles, and of one update more than a tile, against the oracle
    (eps = 1: exact trajectories whatever the table holds; lr = 0.5 so that folded groups show)."""
    seed, id0, eps, lr, gamma, steps = 5, 3, 1.0, 0.5, 0.9, 12
    env = pkg.BatchedGame2048Env(B, seed=seed, env_id0=id0, device=DEV)
    agent = pkg.BatchedQLearningAgent(100, learning_rate=lr, discount_factor=gamma, exploration_rate=eps,
                                      capacity_log2=18, seed=seed, env_id0=id0, device=DEV)
    agent.deterministic_rollout(env, steps)
    envs = O.envs_init(B, 4, seed, id0)
    oa = O.Agent(100, 4, lr, gamma, eps)
    O.rollout_sync(envs, oa, steps, seed, id0, 0)
    assert np.array_equal(env.boards.cpu().numpy(), envs["board"][:, :16])
    keys, vals = oa.dump()
    got = agent.q_values(t8(keys)).cpu().numpy()
    assert np.allclose(got, vals, rtol=1e-5, atol=1e-6)
    assert agent.table_size() == len(oa) and agent.check_status() == 0


# ---------------------------------------------------------------------------------------------
# table placement probe
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [4, 5])
def test_table_probe_leaves_a_live_table_untouched(pkg, n):
    """q2048_table_probe issues scattered atomic ORs of 0: every byte of a populated table is
    as before."""
    env = pkg.BatchedGame2048Env(4096, board_size=n, seed=3, device=DEV)
    agent = pkg.BatchedQLearningAgent(100, learning_rate=0.1, discount_factor=0.9, exploration_rate=0.5,
                                      capacity_log2=16, seed=3, device=DEV, board_size=n)
    assert agent.placement == {"mode": "plain"}          # "auto": small tables are plain
    agent.fused_rollout(env, 10)
    assert agent.table_size() > 4096
    before = agent.table.clone()
    L = LIB(pkg)
    for seed in (0, 1):
        assert L.q2048_table_probe(agent.table.data_ptr(), 16, 1 << 16, 8, seed, None) == 0
    sync()
    assert torch.equal(agent.table, before)
    assert L.q2048_table_probe(None, 16, 8, 8, 0, None) == -1
    assert L.q2048_table_probe(agent.table.data_ptr(), 16, -1, 8, 0, None) == -2


def test_table_placements_give_the_same_learner(pkg):
    """A plain table, the best of three probed candidates and a table mapped from 2 MiB physical
    chunks (q2048_table_alloc): zeroed on arrival, same rollout; the chunked table's memory goes back
    to the device when the agent does; auto_capacity_log2 honours the load bound and the budget."""
    ref = None
    free0, _ = torch.cuda.mem_get_info(torch.device(DEV))
    for placement in ("plain", 3, "chunks"):
        env = pkg.BatchedGame2048Env(8192, seed=5, device=DEV)
        agent = pkg.BatchedQLearningAgent(100, learning_rate=0.1, discount_factor=0.9, exploration_rate=1.0,
                                          capacity_log2=21, seed=5, device=DEV, placement=placement)
        assert agent.table.shape == (1 << 21, 32) and int(agent.table.max()) == 0
        rep = agent.placement
        if placement == 3:
            assert rep["mode"] == "candidates" and rep["candidates"] == 3 and len(rep["probe_us"]) == 3
            # (the report rounds the times: equal to the minimum, not necessarily its first index)
            assert rep["probe_us"][rep["chosen"]] == min(rep["probe_us"]) > 0
        if placement == "chunks":
            assert rep["mode"] == "chunks" and min(rep["probe_us"]) > 0 and agent.table.data_ptr() % (2 << 20) == 0
            assert rep["candidates"] == len(rep["probe_us"]) == 4 and rep["probe_us"][rep["chosen"]] == min(rep["probe_us"])
        agent.fused_rollout(env, 24)           # eps = 1: trajectories do not depend on Q
        keys, q = agent.export_rows()
        order = np.argsort(keys)
        got = (env.boards.cpu().numpy().copy(), keys[order], q[order], agent.stats()["inserts"])
        if ref is None:
            ref = got
        else:
            assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
            assert got[3] == ref[3]
            # rows no two lanes shared carry exactly the same values (about 1 % of the rows are the
            # opening states many lanes race on: those may differ from run to run)
            same = np.isclose(got[2], ref[2], rtol=1e-5, atol=1e-6).all(axis=1)
            assert same.mean() > 0.97
        del agent, env
    release_cached_device_memory()
    assert torch.cuda.mem_get_info(torch.device(DEV))[0] >= free0 - (1 << 30)       # nothing big stayed behind
    # "auto": a 1 GiB table comes in chunks (a 64 MiB one plain); of its four candidates only the winner is
    # still mapped when place_table returns, and its memory goes back when the tensor does
    import gc
    sync()
    before = torch.cuda.mem_get_info(torch.device(DEV))[0]
    t, rep = pkg.place_table(25, torch.device(DEV))
    assert rep["mode"] == "chunks" and rep["candidates"] == 4 and t.shape == (1 << 25, 32) and int(t[::4097].max()) == 0
    sync()
    held = before - torch.cuda.mem_get_info(torch.device(DEV))[0]
    assert (1 << 30) <= held < (1 << 30) + (256 << 20), held                        # the three losers were released
    del t
    gc.collect()
    sync()
    assert before - torch.cuda.mem_get_info(torch.device(DEV))[0] < (128 << 20)
    L = LIB(pkg)
    out = C.c_void_p()
    assert L.q2048_table_alloc(3, 0, C.byref(out)) == -2 and L.q2048_table_alloc(20, 12345, C.byref(out)) == -2
    assert L.q2048_table_alloc(20, 0, None) == -1 and L.q2048_table_free(None) == 0
    assert L.q2048_table_free(0x1000) == -1                                          # not one of ours
    with pytest.raises(ValueError):
        pkg.place_table(16, torch.device(DEV), placement=0)
    free, _ = torch.cuda.mem_get_info(torch.device(DEV))
    cap = pkg.auto_capacity_log2(1000, DEV)
    assert cap == 30 or 32 << cap <= 0.25 * free < 32 << (cap + 1)       # the 32 GiB floor, memory allowing
    assert pkg.auto_capacity_log2(1000, DEV, max_log2=22) == 22
    assert pkg.auto_capacity_log2(1000, DEV, floor_log2=0) == 20
    assert pkg.auto_capacity_log2(1 << 26, DEV, max_log2=22) == 27       # the load bound wins
    assert pkg.auto_capacity_log2(1 << 31, DEV) == 32


def test_chunked_tables_survive_reallocation(pkg):
    """Six 4 GiB tables from q2048_table_alloc one after the other, 4x4 and 5x5 in turn, each freed
    before the next is made: every one arrives zero-filled, ends with exactly as many occupied slots as
    rows were created, finds all its current states, and gets addresses no earlier table had.  (With the
    address range returned to the runtime and handed out again, the third table of a process lost 1-15 %
    of its rows on ROCm 7.2 -- translations of the range's previous life -- so q2048_table_free keeps
    the range reserved: csrc `release_chunks`.)"""
    import gc

    seen, want = set(), {}
    for it in range(6):
        n = 5 if it % 2 else 4
        env = pkg.BatchedGame2048Env(1 << 19, board_size=n, seed=31, env_id0=11, device=DEV)
        agent = pkg.BatchedQLearningAgent(1000, learning_rate=0.1, discount_factor=0.99, exploration_rate=0.2,
                                          capacity_log2=27, seed=31, env_id0=11, device=DEV, board_size=n,
                                          independent=True, placement="chunks")
        assert agent.table.data_ptr() not in seen and int((agent.table.view(torch.int64) != 0).sum()) == 0
        seen.add(agent.table.data_ptr())
        for _ in range(2):
            agent.fused_rollout(env, 16)
        st = agent.stats()
        _, found = agent.q_values(env.boards[:8192], return_found=True)
        assert st["inserts"] == agent.table_size() and st["drops"] == 0 and bool(found.all()), (it, n)
        assert want.setdefault(n, st["inserts"]) == st["inserts"]          # private rows: the same run every time
        assert agent.check_status() == 0
        del agent, env, found
        gc.collect()
    assert pkg._native.claim_timeouts() == 0


@pytest.mark.parametrize("n", [4, 5])
def test_table_grows_like_the_reference_defaultdict(pkg, n):
    """capacity_log2="auto": the reference's q_table is a defaultdict with no capacity
    (Agent/main.py:16); here the table starts small and DOUBLES between launches whenever half of it
    is in use (q2048_table_grow: the next capacity mapped onto fresh chunks in an address range of its own,
    rows moved by one streaming kernel, the smaller table released).  262 144 envs with private rows at
    eps = 0.2 (actions depend on Q), driven until the table has doubled at least twice: boards, aux,
    the key set and EVERY Q row equal, bit for bit, the same job on a table of fixed capacity;
    inserts == len(q_table); nothing dropped; the run-time self-check passes after every launch; every
    table address is new.  Then a shared table at eps = 1 (the key set is a function of the draws
    alone) through the 4-call API and the deterministic step as well, a checkpoint round trip through
    a growing import, and the argument errors of the C entry points."""
    B, S, launches, seed, id0 = 1 << 18, 8, 6 if n == 4 else 4, 5, 77

    def mk(cap, **kw):
        e = pkg.BatchedGame2048Env(B, board_size=n, seed=seed, env_id0=id0, device=DEV)
        a = pkg.BatchedQLearningAgent(1000, learning_rate=0.1, discount_factor=0.95, exploration_rate=0.2,
                                      capacity_log2=cap, seed=seed, env_id0=id0, device=DEV, board_size=n,
                                      independent=True, **kw)
        return e, a

    def table(a):
        k, q = a.export_rows()
        o = np.lexsort(k.reshape(len(q), -1).T[::-1])
        return k[o], q[o]

    e0, a0 = mk(26)
    # the growth off the critical path (round 5: q2048_table_grow_begin / _commit / _finish; the next table mapped
    # by the library's host thread while launches go on) in doubling steps, the same in the default fourfold steps
    # without prefetch, and round 4's host-synchronous q2048_table_grow: one result
    e1, a1 = mk("auto", initial_capacity_log2=21, growth_step_log2=1)
    e3, a3 = mk("auto", initial_capacity_log2=21, growth_step_log2=1, async_growth=False)
    e4, a4 = mk("auto", initial_capacity_log2=21, prefetch_growth=False, verify_growth=True)
    for a in (a1, a3, a4):
        assert a.growable and a.capacity_log2 == 21 and a.placement["mode"] == "chunks"
    assert a1._growth is not None and a3._growth is None and a4._growth is None      # prefetch: begun at construction
    seen = {a1.table.data_ptr()}
    for _ in range(launches):
        a0.fused_rollout(e0, S)
        for e, a in ((e1, a1), (e3, a3), (e4, a4)):
            a.fused_rollout(e, S)
            chk = a.verify_table()                                # occupied slots == rows created, every launch
            assert chk["load"] <= 0.85                            # (soft limit 0.5, hard limit 0.75 + one launch)
        if a1.table.data_ptr() not in seen:
            seen.add(a1.table.data_ptr())
    assert len(a1.growths) >= 2 and 2 <= len(seen) <= len(a1.growths) + 1, a1.growths   # (a call may double twice)
    assert [g["to_log2"] - g["from_log2"] for g in a1.growths] == [1] * len(a1.growths)
    assert [g["to_log2"] - g["from_log2"] for g in a3.growths] == [1] * len(a3.growths) and len(a3.growths) >= 2
    assert a4.growths and all(1 <= g["to_log2"] - g["from_log2"] <= 2 for g in a4.growths)
    assert all(g["rows"] == g["expected_rows"] for a in (a1, a3, a4) for g in a.growths)
    (k0, q0), (k1, q1) = table(a0), table(a1)
    assert np.array_equal(k0, k1) and np.array_equal(q0, q1)      # every row, bit for bit
    for e, a in ((e1, a1), (e3, a3), (e4, a4)):
        assert torch.equal(e0.boards, e.boards) and torch.equal(e0.aux, e.aux)
        k, q = table(a)
        assert np.array_equal(k0, k) and np.array_equal(q0, q)
    s0, s1 = a0.stats(), a1.stats()
    assert s0["inserts"] == s1["inserts"] == len(q1) == a1.table_size() and s1["drops"] == s0["drops"] == 0
    assert a1.check_status() == 0 and pkg._native.claim_timeouts() == 0
    print(f"[grow {n}x{n}] async {a1.growths}\n  sync {a3.growths}\n  fourfold, no prefetch, counted {a4.growths}")
    # the C entry points of the asynchronous growth: one growth per table, abort gives the prepared table back
    L = LIB(pkg)
    own, g1, g2, out = a4.table._q2048_owner, C.c_void_p(), C.c_void_p(), C.c_void_p()
    if a4._growth is not None:
        a4._growth.abort()
        a4._growth = None
    a4.finish_growth()
    if a4.capacity_log2 < a4.max_capacity_log2:
        assert L.q2048_table_grow_begin(own.ptr, a4.capacity_log2, a4.capacity_log2 + 1, C.byref(g1)) == 0
        assert L.q2048_table_grow_begin(own.ptr, a4.capacity_log2, a4.capacity_log2 + 1, C.byref(g2)) == pkg._native.ERR_BUSY
        assert L.q2048_table_grow_finish(g1, None) == -1          # not committed
        assert L.q2048_table_grow_commit(g1, 3, 0, C.byref(out), None) == -2 and L.q2048_table_grow_commit(g1, 1, 8, C.byref(out), None) == -7
        assert L.q2048_table_grow_abort(g1) == 0 and L.q2048_table_grow_abort(g1) == -1 and L.q2048_table_grow_poll(g1) == -1
        assert L.q2048_table_grow_begin(own.ptr, a4.capacity_log2, a4.max_capacity_log2 + 1, C.byref(g1)) == -2
    assert a4.verify_table()["rows"] == len(q1)
    del a3, a4, e3, e4
    # a fixed table cannot grow; a growable one not beyond its range
    with pytest.raises(RuntimeError):
        a0.grow_table()
    with pytest.raises(ValueError):
        a1.grow_table(a1.max_capacity_log2 + 1)
    # checkpoint round trip: rows imported into a SMALLER growable table make it grow first
    sd = a1.state_dict()
    e2, a2 = mk("auto", initial_capacity_log2=18)
    a2.load_state_dict(sd)
    assert a2.capacity_log2 >= 18 + 2 and a2.table_size() == len(q1) and a2.verify_table()["rows"] == len(q1)
    k2, q2 = table(a2)
    assert np.array_equal(k2, k1) and np.array_equal(q2, q1)
    del a0, a1, a2, e0, e1, e2, sd
    # shared table, eps = 1: fused launches, 4-call iterations and deterministic steps on one growing table
    Bs = 1 << 16
    es = pkg.BatchedGame2048Env(Bs, board_size=n, seed=seed, env_id0=id0, device=DEV)
    ag = pkg.BatchedQLearningAgent(1000, exploration_rate=1.0, capacity_log2="auto", initial_capacity_log2=17,
                                   growth_step_log2=1, seed=seed, env_id0=id0, device=DEV, board_size=n)
    ef = pkg.BatchedGame2048Env(Bs, board_size=n, seed=seed, env_id0=id0, device=DEV)
    af = pkg.BatchedQLearningAgent(1000, exploration_rate=1.0, capacity_log2=24, seed=seed, env_id0=id0,
                                   device=DEV, board_size=n)
    for e, a in ((es, ag), (ef, af)):
        a.fused_rollout(e, 6)
        s = e.boards
        for _ in range(4):
            act = a.choose_action(s)
            s2, r, d, _ = e.step(act)
            a.update_q_value(s, act, r, s2, d)
            s = e.reset(d)
        a.deterministic_rollout(e, 4)
        a.fused_rollout(e, 6)
    ag.finish_growth()
    assert len(ag.growths) >= 3 and torch.equal(es.boards, ef.boards)
    kg, kf = ag.export_rows()[0], af.export_rows()[0]
    kg, kf = kg.reshape(len(kg), -1), kf.reshape(len(kf), -1)
    assert np.array_equal(kg[np.lexsort(kg.T[::-1])], kf[np.lexsort(kf.T[::-1])])   # the same key set
    assert ag.verify_table()["rows"] == len(kf) == ag.stats()["inserts"] and ag.stats()["drops"] == 0
    # the C entry points' argument errors
    L = LIB(pkg)
    out, moved = C.c_void_p(), C.c_int64()
    own = ag.table._q2048_owner
    assert L.q2048_table_grow(own.ptr, ag.capacity_log2 - 1, ag.capacity_log2 + 1, 1, C.byref(out), C.byref(moved), None) == -2
    assert L.q2048_table_grow(own.ptr, ag.capacity_log2, ag.capacity_log2, 1, C.byref(out), C.byref(moved), None) == -2
    assert L.q2048_table_grow(own.ptr, ag.capacity_log2, own.max_capacity_log2 + 1, 1, C.byref(out), C.byref(moved), None) == -2
    assert L.q2048_table_grow(own.ptr, ag.capacity_log2, ag.capacity_log2 + 1, 3, C.byref(out), C.byref(moved), None) == -2
    assert L.q2048_table_grow(af.table.data_ptr(), 24, 25, 1, C.byref(out), C.byref(moved), None) == (
        -2 if getattr(af.table, "_q2048_owner", None) is not None else -1)      # a family of one / not the library's
    assert L.q2048_table_reserve(20, 19, 0, C.byref(out)) == -2 and L.q2048_table_reserve(20, 22, 12345, C.byref(out)) == -2


def test_growth_commit_contract_and_two_thread_race(pkg):
    """q2048_table_grow_commit's error contract and its critical section, through the
    C ABI alone:
      * begin(A) -> commit -> begin(B) -> commit(B) BEFORE finish(A): Q2048_ERR_BUSY, and the growth of B is STILL
        VALID (poll answers, the prepared table is not leaked): after finish(A) the same handle commits;
      * the host-synchronous q2048_table_grow on a table whose family has an unfinished commit: Q2048_ERR_BUSY, and
        nothing stays behind -- the next begin on that table works (round 5 left it BUSY until q2048_table_free);
      * two host threads committing ONE growth while its table is still being mapped: exactly one succeeds, the
        other gets Q2048_ERR_BUSY (or, had it arrived after the first was done, Q2048_ERR_NULL); the rows are moved once."""
    import threading

    N, L = pkg._native, LIB(pkg)
    release_cached_device_memory()                        # (the family below reserves room for a 32 GiB table)
    env = pkg.BatchedGame2048Env(4096, seed=2, device=DEV)
    agent = pkg.BatchedQLearningAgent(100, exploration_rate=1.0, capacity_log2="auto", initial_capacity_log2=18,
                                      max_capacity_log2=30, prefetch_growth=False, load_limit=0.9, seed=2, device=DEV,
                                      freeze_load=None)
    agent.fused_rollout(env, 20)
    rows = agent.verify_table()["rows"]
    assert agent.capacity_log2 == 18 and not agent.growths and agent._growth is None and rows > 4096
    assert agent.max_capacity_log2 == 30
    a_ptr, stream = agent.table._q2048_owner.ptr, None
    g1, g2, b_ptr, c_ptr, moved = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int64()
    sync()
    assert L.q2048_table_grow_begin(a_ptr, 18, 19, C.byref(g1)) == 0
    assert L.q2048_table_grow_commit(g1, 1, 0, C.byref(b_ptr), stream) == 0
    assert L.q2048_table_grow_begin(b_ptr, 19, 20, C.byref(g2)) == 0             # allowed right after the commit
    assert L.q2048_table_grow_commit(g2, 1, 0, C.byref(c_ptr), stream) == N.ERR_BUSY and not c_ptr.value
    assert L.q2048_table_grow_wait(g2, None) == 0 and L.q2048_table_grow_poll(g2) == 0    # g2 is alive and prepared
    g3, d_ptr = C.c_void_p(), C.c_void_p()
    assert L.q2048_table_grow_begin(b_ptr, 19, 21, C.byref(g3)) == N.ERR_BUSY    # one growth per table: g2
    assert L.q2048_table_grow_finish(g1, C.byref(moved)) == 0 and moved.value == rows
    assert L.q2048_table_grow_commit(g2, 1, 0, C.byref(c_ptr), stream) == 0 and c_ptr.value
    # the synchronous form on C while g2 is committed and unfinished: BUSY, and nothing of it stays behind
    assert L.q2048_table_grow(c_ptr, 20, 21, 1, C.byref(d_ptr), C.byref(moved), stream) == N.ERR_BUSY
    assert L.q2048_table_grow_finish(g2, C.byref(moved)) == 0 and moved.value == rows
    assert L.q2048_table_grow(c_ptr, 20, 21, 1, C.byref(d_ptr), C.byref(moved), stream) == 0 and moved.value == rows
    # two threads, one growth: 2^21 -> 2^30 slots (32 GiB: tens of milliseconds of mapping, both threads arrive meanwhile)
    g4, e_ptr = C.c_void_p(), [C.c_void_p(), C.c_void_p()]
    assert L.q2048_table_grow_begin(d_ptr, 21, 30, C.byref(g4)) == 0
    codes = [None, None]

    def commit(k):
        with torch.cuda.device(torch.device(DEV)):
            codes[k] = L.q2048_table_grow_commit(g4, 1, 0, C.byref(e_ptr[k]), stream)

    threads = [threading.Thread(target=commit, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert sorted(codes)[1] == 0 and sorted(codes)[0] in (N.ERR_BUSY, -1), codes
    winner = codes.index(0)
    assert e_ptr[winner].value and not e_ptr[1 - winner].value
    assert L.q2048_table_grow_finish(g4, C.byref(moved)) == 0 and moved.value == rows      # moved once, every row
    count = torch.zeros(1, dtype=torch.int64, device=DEV)
    assert L.q2048_table_count(e_ptr[winner], 30, count.data_ptr(), stream) == 0
    assert int(count.item()) == rows
    assert L.q2048_table_free(e_ptr[winner]) == 0          # the family's last live table: its retired ones, stream and scratch go too
    agent.table._q2048_owner._finalizer.detach()           # (A was retired by finish(g1) and released with its family)
    del agent
    print(f"[commit race] codes {codes}: one commit, {rows} rows moved once")


def test_growth_preparation_really_fails_when_the_device_has_no_room(pkg):
    """the failed-preparation path was only ever exercised through a monkeypatched commit.  Here the device
    really has no room: a family that may grow to 2^31 slots (64 GiB), all but ~12 GiB of the device taken by another
    allocation, begin(2^18 -> 2^31): the library's host thread finds no room -- it ASKS (hipMemGetInfo) before it creates
    a single chunk -- and reports Q2048_ERR_ALLOC through wait / poll; commit returns the same code and ends the growth
    with the old table intact (every row still there), and the table is not left "in a growth": a growth that fits
    (2^19 slots) begins, is prepared and aborts right away.
    (The first version of this test let the library map chunk after chunk until hipMemCreate failed.  That failed
    cleanly, every chunk came back -- and the process's next large mapping ended in a GPU memory fault during its zero
    fill, twice in two runs: profiles/r06_va_reuse_fault.txt.  A device is never walked into exhaustion any more, by
    the library or by this test.)"""
    if DEV == "cpu":
        pytest.skip("the chunk allocator is the device library's")
    N, L = pkg._native, LIB(pkg)
    release_cached_device_memory()
    env = pkg.BatchedGame2048Env(4096, seed=4, device=DEV)
    agent = pkg.BatchedQLearningAgent(100, exploration_rate=1.0, capacity_log2="auto", initial_capacity_log2=18,
                                      max_capacity_log2=31, prefetch_growth=False, load_limit=0.9, seed=4, device=DEV,
                                      freeze_load=None)
    agent.fused_rollout(env, 20)
    rows = agent.verify_table()["rows"]
    assert agent.max_capacity_log2 == 31 and agent._growth is None
    a_ptr = agent.table._q2048_owner.ptr
    assert 0x100000000000 <= a_ptr < 0x600000000000                                  # tables live in a private region (16 TiB upward)
    free, _ = torch.cuda.mem_get_info(torch.device(DEV))
    hog = torch.empty(int(free - (12 << 30)), dtype=torch.uint8, device=DEV)        # leaves ~12 GiB of the device
    left, _ = torch.cuda.mem_get_info(torch.device(DEV))
    g, out, ms = C.c_void_p(), C.c_void_p(), C.c_double()
    assert L.q2048_table_grow_begin(a_ptr, 18, 31, C.byref(g)) == 0
    assert L.q2048_table_grow_wait(g, C.byref(ms)) == N.ERR_ALLOC and L.q2048_table_grow_poll(g) == N.ERR_ALLOC
    assert L.q2048_table_grow_commit(g, 1, 0, C.byref(out), None) == N.ERR_ALLOC and not out.value
    assert L.q2048_table_grow_poll(g) == -1                                          # the growth is gone
    left_after, _ = torch.cuda.mem_get_info(torch.device(DEV))
    assert left_after >= left - (64 << 20)                                           # nothing of the refused table is held
    assert agent.table_size() == rows and agent.verify_table()["rows"] == rows       # the old table is intact
    assert L.q2048_table_grow_begin(a_ptr, 18, 19, C.byref(g)) == 0                  # ... and not stuck in a growth
    assert L.q2048_table_grow_wait(g, C.byref(ms)) == 0 and L.q2048_table_grow_abort(g) == 0
    agent.fused_rollout(env, 5)
    assert agent.verify_table()["rows"] > rows and agent.check_status() == 0
    del hog
    release_cached_device_memory()


def test_growing_table_falls_back_to_a_fixed_one_when_it_cannot_be_mapped(pkg, monkeypatch):
    """`capacity_log2="auto"` on a stack without the virtual-memory API, or without room for the first
    table, used to raise out of the constructor (and `train.py`'s default with it).  The chunk allocator is made to
    fail here: the agent warns, takes ONE fixed plain table and trains on it; a table that is not growable refuses
    `grow_table`."""
    agent_mod = __import__("importlib").import_module("2048_q-learning_amd.agent")

    def refuse(self, *a, **kw):
        raise pkg._native.NativeError("q2048_table_reserve: device memory could not be reserved, created or mapped (code -8)")

    monkeypatch.setattr(agent_mod._ChunkedTable, "__init__", refuse)
    monkeypatch.setattr(agent_mod, "auto_capacity_log2", lambda *a, **kw: 22)          # (keep the fallback table small)
    with pytest.warns(UserWarning, match="could not be mapped"):
        agent = pkg.BatchedQLearningAgent(100, exploration_rate=0.5, capacity_log2="auto", seed=2, device=DEV)
    assert not agent.growable and agent.capacity_log2 == 22 and agent.placement["mode"] == "plain"
    assert "growable_failed" in agent.placement and agent._growth is None
    env = pkg.BatchedGame2048Env(4096, seed=2, device=DEV)
    agent.fused_rollout(env, 32)
    st = agent.stats()
    assert st["steps"] == 4096 * 32 and st["inserts"] == agent.table_size() > 0 and st["drops"] == 0
    with pytest.raises(RuntimeError):
        agent.grow_table()


def test_growth_that_finds_no_room_caps_the_table_instead_of_ending_the_run(pkg, monkeypatch):
    """The device has no room for the next table (another tenant, a smaller card): the preparation fails, the commit
    reports Q2048_ERR_ALLOC, and the run goes on on the table it has -- that capacity is the largest from then on, a
    warning says so -- instead of an exception out of `fused_rollout` in mid-training."""
    agent_mod = __import__("importlib").import_module("2048_q-learning_amd.agent")
    N = pkg._native
    real_commit = agent_mod._Growth.commit

    def no_room(self, *a, **kw):
        self.abort()                                       # (give the prepared table back: the failure is simulated)
        raise N.NativeError("q2048_table_grow_commit: device memory could not be reserved, created or mapped (code -8)", N.ERR_ALLOC)

    env = pkg.BatchedGame2048Env(1 << 16, seed=4, device=DEV)
    agent = pkg.BatchedQLearningAgent(100, exploration_rate=1.0, capacity_log2="auto", initial_capacity_log2=18,
                                      max_capacity_log2=24, seed=4, device=DEV)
    monkeypatch.setattr(agent_mod._Growth, "commit", no_room)
    with pytest.warns(UserWarning, match="cannot grow beyond"):
        for _ in range(4):
            agent.fused_rollout(env, 8)
    assert agent.capacity_log2 == 18 == agent.max_capacity_log2 and agent._growth is None and not agent.growths
    monkeypatch.setattr(agent_mod._Growth, "commit", real_commit)
    agent.fused_rollout(env, 8)                            # and on it goes, on the table it has
    st = agent.stats()
    assert st["steps"] == (1 << 16) * 40 and agent.table_size() == st["inserts"]
    assert agent.verify_table()["capacity_log2"] == 18


def test_process_exit_with_a_growth_in_flight(tmp_path):
    """A process may end at any point of a growth: while the library's host thread is still mapping the next table
    (prefetch begun, never waited for), and with a committed growth nobody finished.  The thread is joined by an
    atexit handler registered after the HIP runtime's own (so it runs first); nothing hangs, nothing crashes."""
    import subprocess
    import sys

    from conftest import REPO

    code = """
import importlib, sys
sys.path.insert(0, %r)
pkg = importlib.import_module("2048_q-learning_amd")
mode = sys.argv[1]
env = pkg.BatchedGame2048Env(65536, seed=1, device="cuda:0")
agent = pkg.BatchedQLearningAgent(100, exploration_rate=1.0, capacity_log2="auto", initial_capacity_log2=22,
                                  max_capacity_log2=28, seed=1, device="cuda:0")
assert agent._growth is not None                      # the 2^24-slot table is being mapped right now
if mode == "committed":
    for _ in range(6):
        agent.fused_rollout(env, 8)                   # passes a quarter of the limit: the growth is committed ...
    assert agent.capacity_log2 > 22 and agent._retiring is not None or agent.growths
print("exiting", mode, flush=True)                    # ... and the process ends without finish / wait / free
""" % REPO
    for mode in ("preparing", "committed"):
        p = subprocess.run([sys.executable, "-c", code, mode], capture_output=True, text=True, timeout=120)
        assert p.returncode == 0 and "exiting " + mode in p.stdout, (mode, p.returncode, p.stderr[-1500:])


def test_device_spanning_table_uses_64_bit_slot_indices(pkg):
    """A 2^32-slot table (128 GiB: what a run of 2^31 rows is given): rows land above slot 2^31 and
    above byte offset 2^36, every insert is a row, no drops."""
    dev = torch.device(DEV)
    release_cached_device_memory()                           # tables cached by earlier tests
    cap = pkg.auto_capacity_log2(1 << 31, dev, max_log2=32)
    free, _ = torch.cuda.mem_get_info(dev)
    if cap < 32 or free < (140 << 30):
        pytest.skip(f"free device memory does not allow 2^32 slots")
    env = pkg.BatchedGame2048Env(1 << 20, seed=11, device=DEV)
    agent = pkg.BatchedQLearningAgent(100, learning_rate=0.1, discount_factor=0.9, exploration_rate=0.9,
                                      capacity_log2=cap, seed=11, device=DEV)
    assert agent.placement == {"mode": "plain"}
    for _ in range(2):
        agent.fused_rollout(env, 16)
    st = agent.stats()
    assert st["drops"] == 0 and agent.check_status() == 0
    assert agent.table_size() == st["inserts"] > (1 << 22)
    upper = agent.table[(1 << 31):, :8]                      # key words of the upper half
    frac_upper = float((upper != 0).any(dim=1).sum()) / st["inserts"]
    assert 0.45 < frac_upper < 0.55                          # uniformly hashed over all 2^32 slots
    q = agent.q_values(env.boards[:4096])                    # current states are in the table
    assert q.shape == (4096, 4) and bool(torch.isfinite(q).all())
    del agent, env, upper
    torch.cuda.empty_cache()


def test_5x5_contended_creation_never_times_out(pkg, O):
    """Many lanes create the SAME 5x5 rows at the same time (every env starts from one of a few
    hundred boards): one compare-and-swap per row, the owner publishes the second key word,
    everybody else waits for it.  No duplicates, no lost rows, and no lane ever gave up waiting --
    here or in any earlier test of this process."""
    B, steps = 1 << 16, 6
    env = pkg.BatchedGame2048Env(B, board_size=5, seed=21, device=DEV)
    agent = pkg.BatchedQLearningAgent(10, exploration_rate=1.0, capacity_log2=22, seed=21, device=DEV,
                                      board_size=5)
    agent.fused_rollout(env, steps)
    envs = O.envs_init(B, 5, 21, 0)
    oa = O.Agent(10, 4, 0.1, 0.9, 1.0, n=5)
    O.rollout(envs, oa, steps, 21, 0, 0)
    assert np.array_equal(env.boards.cpu().numpy(), envs["board"][:, :25])
    st = agent.stats()
    assert st["drops"] == 0 and agent.table_size() == st["inserts"] == len(oa)   # the dict's key set size
    keys, _ = agent.export_rows()
    assert len(np.unique(keys, axis=0)) == len(keys)                              # no duplicate rows
    assert pkg._native.claim_timeouts() == 0


# ---------------------------------------------------------------------------------------------
# multi-rank rehearsal on the one leased GPU
# ---------------------------------------------------------------------------------------------
def test_bench_self_launched_two_ranks_share_one_gpu():
    """`python bench.py --gpus 2` with no process group in the environment: the script starts its
    own two ranks (Q2048_DIST_BACKEND=gloo lets both sit on the one GPU of this box), each owns
    half of the global env ids and its own table replica, and rank 0 prints ONE line whose
    statistics are the all-reduced whole-job numbers.  (RCCL itself needs two GPUs: never run here.)"""
    release_cached_device_memory()
    import subprocess
    import sys

    from conftest import REPO

    B, K = 1 << 16, 24
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["Q2048_DIST_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", str(K),
                        "--warmup", "8", "--boards-per-gpu", str(B), "--cap-log2", "24", "--prep-steps", "256",
                        "--repeats", "3", "--cpu-seconds", "0"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["steps"] == K and rec["warmup"] == 8
    assert rec["config"]["total_boards"] == 2 * B and rec["config"]["boards_per_gpu"] == B
    assert rec["stats"]["episodes"] > 0 and rec["stats"]["drops"] == 0 and rec["stats"]["status"] == 0
    assert rec["cpu_baseline"] is None and "companions" not in rec          # --cpu-seconds 0
    assert len(rec["region_ms"]) == 3 and rec["value"] > 0
    assert abs(rec["value"] - 2 * B * K / (rec["ms_per_step"] * K / 1e3)) < 1e-6 * rec["value"]
    # the CPU path is timed beside the GPU number at N > 1 too (rank 0, after the GPU regions)
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "8",
                        "--warmup", "4", "--boards-per-gpu", str(1 << 14), "--cap-log2", "22", "--prep-steps", "256",
                        "--repeats", "2", "--cpu-seconds", "1.5"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][0])
    assert rec["n_gpus"] == 2 and rec["cpu_baseline"]["kind"] == "port" and rec["cpu_baseline"]["value"] > 1e5
    assert rec["cpu_baseline"]["cores"] >= 1 and "single_thread" in rec["cpu_baseline"]


def test_check_shards_across_processes():
    """World-size invariance proved ACROSS PROCESSES: `bench.py --check-shards` with one rank on
    32 768 boards and with two self-launched ranks (gloo; both on this box's one GPU) on 16 384 boards
    each -- the same global env ids 0..32767 -- prints the same 64-bit hash of boards + aux for every
    chunk of 4096 ids after 300 steps (episodes end and reset on the way); another seed does not."""
    release_cached_device_memory()
    import subprocess
    import sys

    from conftest import REPO

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["Q2048_DIST_BACKEND"] = "gloo"

    def run(gpus, per_gpu, seed=0, n=4):
        p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(gpus), "--check-shards",
                            "--steps", "300", "--boards-per-gpu", str(per_gpu), "--seed", str(seed),
                            "--board-size", str(n)], capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
        assert len(lines) == 1
        return json.loads(lines[0])

    one, two = run(1, 32768), run(2, 16384)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and one["total_envs"] == two["total_envs"] == 32768
    assert len(one["hashes"]) == 8 and one["hashes"] == two["hashes"] and len(set(one["hashes"])) == 8
    assert one["episodes"] == two["episodes"] > 1000 and one["env_steps"] == two["env_steps"] == 32768 * 300
    assert one["status"] == two["status"] == 0
    assert run(1, 32768, seed=1)["hashes"] != one["hashes"]
    five1, five2 = run(1, 8192, n=5), run(2, 4096, n=5)
    assert five1["hashes"] == five2["hashes"] and len(five1["hashes"]) == 2


def test_bench_under_torchrun_one_rank_drives_rccl():
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`: a launched rank joins
    its process group even when it is alone, so this is the N > 1 code path end to end on RCCL --
    `init_process_group("nccl", device_id=...)`, the statistics all-reduce on the side stream, the MAX
    all-reduce of the region times, the barriers -- with the one rank a one-GPU box has."""
    release_cached_device_memory()
    import socket
    import subprocess
    import sys

    from conftest import REPO

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "Q2048_DIST_BACKEND")}
    B, K = 1 << 16, 16
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", str(K), "--warmup", "4",
                        "--boards-per-gpu", str(B), "--cap-log2", "24", "--prep-steps", "128", "--repeats", "3",
                        "--cpu-seconds", "0", "--no-companions"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    rec = json.loads([ln for ln in p.stdout.splitlines() if ln.lstrip().startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["config"]["total_boards"] == B and rec["stats"]["episodes"] > 0
    assert "RCCL" in rec["config"]["parallelism"] and rec["stats"]["status"] == 0


@pytest.mark.gpu
def test_stats_reduction_on_the_device_is_a_snapshot(pkg):
    """StatsAllReduce with one process: the vectors reach the host by one copy when they are the two
    views of an agent's buffer, by two when they are separate tensors -- either way what `wait` returns
    is what the vectors held when `start` was called, whatever is launched after it."""
    N = pkg._native
    r = pkg.StatsAllReduce(DEV)
    agent = pkg.BatchedQLearningAgent(10, capacity_log2=10, seed=1, device=DEV)
    assert pkg.dist._packed(agent.stats_i, agent.stats_f) is not None        # one copy
    for si, sf in ((agent.stats_i, agent.stats_f),
                   (torch.zeros(N.NSTAT_I, dtype=torch.int64, device=DEV),
                    torch.zeros(N.NSTAT_F, dtype=torch.float64, device=DEV))):
        assert (pkg.dist._packed(si, sf) is None) == (si is not agent.stats_i)
        si.copy_(torch.arange(N.NSTAT_I, dtype=torch.int64))
        sf.fill_(2.5)
        r.start(si, sf)
        si.add_(1000)                                      # after the snapshot, on the same stream
        sf.zero_()
        a, b = r.wait()
        assert a.tolist() == list(range(N.NSTAT_I)) and b.tolist() == [2.5] * N.NSTAT_F
        assert int(si[3]) == 1003
    with pytest.raises(RuntimeError):
        r.wait()


# ---------------------------------------------------------------------------------------------
# evaluation of a trained table (the reference README's evaluate.py / models/)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [4, 5])
def test_no_learn_rollout_reads_but_never_writes(pkg, O, n):
    """Q2048_FLAG_NO_LEARN: after some training, play on with the stored values only.  The table is
    bit-identical afterwards (no row created, no value written) and every trajectory equals the
    oracle agent's with its learning rate set to 0 (same lookups, same argmax, frozen values)."""
    B, train_steps, eval_steps, seed, id0, eps = 128, 300, 250, 9, 4242, 0.15
    envs = O.envs_init(B, n, seed, id0)
    agents = [O.Agent(100, 4, 0.1, 0.95, eps, n=n) for _ in range(B)]
    for i in range(B):
        O.rollout(envs[i:i + 1], agents[i], train_steps, seed, id0 + i, 0)
        agents[i]._view().lr = 0.0
        O.rollout(envs[i:i + 1], agents[i], eval_steps, seed, id0 + i, train_steps)
    env = pkg.BatchedGame2048Env(B, board_size=n, seed=seed, env_id0=id0, device=DEV)
    agent = pkg.BatchedQLearningAgent(100, learning_rate=0.1, discount_factor=0.95, exploration_rate=eps,
                                      capacity_log2=18, seed=seed, env_id0=id0, device=DEV,
                                      independent=True, board_size=n)
    agent.fused_rollout(env, train_steps)
    before = agent.table.clone()
    rows = agent.table_size()
    agent.stats(reset=True)
    agent.fused_rollout(env, 100, learn=False)
    agent.fused_rollout(env, eval_steps - 100, learn=False)
    assert torch.equal(agent.table, before) and agent.table_size() == rows       # nothing written
    assert np.array_equal(env.boards.cpu().numpy(), envs["board"][:, :n * n])
    f = env.aux_fields()
    for k, v in oracle_aux(envs).items():
        assert np.array_equal(f[k].astype(np.int64), np.asarray(v, dtype=np.int64)), k
    # the running return is a float32 sum on the device (550 terms here): 1e-4, not the 1e-5 of a Q value
    assert np.allclose(f["ep_return"], envs["episode_return"], rtol=1e-4, atol=1e-3)
    st = agent.stats()
    assert st["steps"] == B * eval_steps and st["inserts"] == 0 and st["drops"] == 0
    assert st["episodes"] > 0 and agent.check_status() == 0


def test_train_save_then_evaluate_scripts(tmp_path):
    """`train.py --save` writes the learner, `evaluate.py` plays it without learning and `train.py --resume`
    continues from it (the README's train / evaluate / models layout).  evaluate.py's two policies: `legal` (argmax
    of the stored row over the moves that change the board: every move but a game's last is valid) and `reference`
    (the lr = 0 agent, argmax over all four actions, Q2048_FLAG_NO_LEARN: a greedy env repeats an invalid argmax
    until the stall rule ends the episode -- faithful, and a far worse player)."""
    release_cached_device_memory()
    import subprocess
    import sys

    from conftest import REPO

    model = str(tmp_path / "models" / "q.pt")
    run = lambda *a: subprocess.run([sys.executable, *a, "--device", DEV], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))  # noqa: E731
    p = run(os.path.join(REPO, "train.py"), "--num-envs", "4096", "--episodes", "3", "--steps-per-launch", "32",
            "--report-every", "4", "--capacity-log2", "24", "--save", model, "--log", str(tmp_path / "t.csv"))
    assert p.returncode == 0 and os.path.exists(model), p.stderr[-2000:]
    recs = {}
    for policy in ("legal", "reference"):
        p = run(os.path.join(REPO, "evaluate.py"), "--model", model, "--num-envs", "2048", "--episodes", "2", "--policy", policy)
        assert p.returncode == 0, p.stderr[-2000:]
        rec = recs[policy] = json.loads(p.stdout.strip().splitlines()[-1])
        # (the table comes from a racing shared-table run: its rows differ a little from run to run)
        assert rec["games"] >= 2 * 2048 and rec["rows"] > 10000 and rec["epsilon"] == 0.0 and rec["policy"] == policy, rec
        assert rec["best_tile"] >= 16 and rec["env_steps"] > 0, rec
    legal, ref = recs["legal"], recs["reference"]
    assert legal["valid_move_frac"] > 0.98 > 0.6 > ref["valid_move_frac"], (legal, ref)
    assert legal["mean_score"] > 5 * ref["mean_score"] and legal["best_tile"] >= 256, (legal, ref)
    assert sum(legal["max_tile_hist"].values()) == legal["games"]
    p = run(os.path.join(REPO, "train.py"), "--num-envs", "4096", "--episodes", "1", "--steps-per-launch", "32",
            "--report-every", "4", "--capacity-log2", "24", "--resume", model, "--log", str(tmp_path / "t2.csv"))
    assert p.returncode == 0, p.stderr[-2000:]


def test_train_resume_continues_the_run(tmp_path):
    """`train.py --stop-epoch k --save` then `--resume ... --episodes N` trains what one N-epoch run
    trains (resume used to restart the epoch counter and replay the decay on an epsilon
    that was already decayed).  Deterministic mode makes the runs reproducible, so the check is
    exact: the resumed run's report rows (epoch, episodes, env-steps, epsilon, mean return, rows)
    are the uninterrupted run's rows from the stop on, and the final tables are bit-identical.
    A resume with a LONGER schedule than the saved run's keeps decaying instead of being pinned at
    epsilon_min by the saved run's limits."""
    release_cached_device_memory()
    import csv
    import subprocess
    import sys

    from conftest import REPO

    common = ["--num-envs", "512", "--steps-per-launch", "32", "--report-every", "1", "--capacity-log2", "22",
              "--deterministic", "--epsilon", "0.9", "--seed", "3"]
    run = lambda *a: subprocess.run([sys.executable, os.path.join(REPO, "train.py"), *common, *a],  # noqa: E731
                                    capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    rows = lambda f: [r[:8] for r in list(csv.reader(open(tmp_path / f)))[1:]]  # noqa: E731  (Drops, Steps/s left out)
    p = run("--episodes", "10", "--save", "full.pt", "--log", "full.csv")
    assert p.returncode == 0, p.stderr[-2000:]
    p = run("--episodes", "10", "--stop-epoch", "4", "--save", "part.pt", "--log", "p1.csv")
    assert p.returncode == 0, p.stderr[-2000:]
    p = run("--episodes", "10", "--resume", "part.pt", "--save", "resumed.pt", "--log", "p2.csv")
    assert p.returncode == 0, p.stderr[-2000:]
    full, p1, p2 = rows("full.csv"), rows("p1.csv"), rows("p2.csv")
    assert int(p1[-1][0]) >= 4 and int(p1[-1][0]) < 10 and len(p2) > 3
    assert p1 + p2 == full                                       # same epochs, episodes, steps, epsilon trace, returns
    assert int(full[-1][0]) == 10 and len({r[3] for r in full}) >= 5          # epsilon really moved
    a, b = (torch.load(tmp_path / f, map_location="cpu", weights_only=False) for f in ("full.pt", "resumed.pt"))
    oa, ob = np.argsort(a["keys"]), np.argsort(b["keys"])
    assert np.array_equal(a["keys"][oa], b["keys"][ob]) and np.array_equal(a["q"][oa], b["q"][ob])
    assert a["train"]["epoch"] == b["train"]["epoch"] == 10 and a["schedule"]["epsilon"] == b["schedule"]["epsilon"]
    # a longer schedule on resume: limits come from the new --episodes
    p = run("--episodes", "40", "--resume", "part.pt", "--stop-epoch", "8", "--log", "p3.csv")
    assert p.returncode == 0, p.stderr[-2000:]
    eps3 = [float(r[3]) for r in rows("p3.csv")]
    # epochs 4..7 of a 40-epoch schedule are its first phase (floor 1.5 * epsilon_min = 0.015, Agent/main.py:46-48);
    # the saved run's own limits (first phase over at epoch 3, last phase from epoch 8) would give 0.011 and 0.01
    assert int(rows("p3.csv")[-1][0]) >= 8 and abs(eps3[-1] - 0.015) < 1e-6


def test_train_default_growing_table_save_and_resume(tmp_path):
    """`train.py` WITHOUT --capacity-log2 (the default since round 4: a table that grows; ADVICE r4: only profile
    logs backed that path).  A small first capacity makes the table grow several times inside the run, off the
    critical path (--growth async, the default) and host-synchronously (--growth sync): in deterministic mode both
    print the same report rows and save bit-identical tables; a save at epoch 3 resumed into a fresh growing table
    ends in the same table as the uninterrupted run; the end-of-run check (occupied slots == rows created) passes."""
    release_cached_device_memory()
    import csv
    import subprocess
    import sys

    from conftest import REPO

    common = ["--num-envs", "2048", "--steps-per-launch", "32", "--report-every", "1", "--initial-capacity-log2", "15",
              "--deterministic", "--epsilon", "0.9", "--seed", "4", "--episodes", "6"]
    run = lambda *a: subprocess.run([sys.executable, os.path.join(REPO, "train.py"), *common, *a],  # noqa: E731
                                    capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    rows = lambda f: [r[:8] for r in list(csv.reader(open(tmp_path / f)))[1:]]  # noqa: E731
    pa = run("--save", "async.pt", "--log", "async.csv")
    assert pa.returncode == 0, pa.stderr[-2000:]
    ps = run("--growth", "sync", "--save", "sync.pt", "--log", "sync.csv")
    assert ps.returncode == 0, ps.stderr[-2000:]
    for p in (pa, ps):
        assert p.stdout.count("table grew") >= 2 and "table check passed" in p.stdout, p.stdout[-1500:]
    assert "host blocked" in pa.stdout and "host-synchronous" in ps.stdout
    assert rows("async.csv") == rows("sync.csv")
    p1 = run("--stop-epoch", "3", "--save", "part.pt", "--log", "p1.csv")
    assert p1.returncode == 0, p1.stderr[-2000:]
    p2 = run("--resume", "part.pt", "--save", "resumed.pt", "--log", "p2.csv")
    assert p2.returncode == 0 and "table check passed" in p2.stdout, p2.stderr[-2000:]
    assert rows("p1.csv") + rows("p2.csv") == rows("async.csv")
    sd = {f: torch.load(tmp_path / f, map_location="cpu", weights_only=False) for f in ("async.pt", "sync.pt", "resumed.pt")}
    o = {f: np.argsort(d["keys"]) for f, d in sd.items()}
    for f in ("sync.pt", "resumed.pt"):
        assert np.array_equal(sd["async.pt"]["keys"][o["async.pt"]], sd[f]["keys"][o[f]])
        assert np.array_equal(sd["async.pt"]["q"][o["async.pt"]], sd[f]["q"][o[f]])
    assert sd["async.pt"]["capacity_log2"] > 15 and len(sd["async.pt"]["keys"]) > (1 << 14)


def test_train_self_launched_two_ranks_share_one_gpu(tmp_path):
    """`python train.py --gpus 2`: two self-launched ranks (gloo, both on this box's one GPU), each
    with its shard of the env ids and its own table; rank 0's CSV holds the all-reduced numbers."""
    release_cached_device_memory()
    import csv
    import subprocess
    import sys

    from conftest import REPO

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["Q2048_DIST_BACKEND"] = "gloo"
    log = tmp_path / "train.csv"
    p = subprocess.run([sys.executable, os.path.join(REPO, "train.py"), "--gpus", "2", "--num-envs", "4096",
                        "--episodes", "2", "--steps-per-launch", "32", "--report-every", "4",
                        "--capacity-log2", "24", "--log", str(log)], capture_output=True, text=True,
                       timeout=600, env=env, cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr[-3000:]
    rows = list(csv.DictReader(open(log)))
    assert rows and int(rows[-1]["Epoch"]) == 2
    assert int(rows[-1]["Episodes"]) >= 2 * 8192                      # both shards' episodes
    assert int(rows[-1]["Env-Steps"]) % (8192 * 32) == 0 and int(rows[-1]["Drops"]) == 0
    # the same job on the default, GROWING tables: every rank grows its own replica off the critical path (its own
    # host thread of the library, its own prefetch) and checks it at the end; epsilon = 1 makes the trajectories
    # independent of the tables, so the all-reduced rows equal the fixed-capacity run's
    log2 = tmp_path / "train_growing.csv"
    # (epsilon_min = 2/3 makes the schedule's first phase flat at 1.0: Agent/main.py:28,46-48)
    common = ["--gpus", "2", "--num-envs", "4096", "--episodes", "1000", "--steps-per-launch", "32", "--report-every", "4",
              "--epsilon", "1.0", "--epsilon-min", "0.6666666666666666", "--max-steps", "256"]
    runs = {}
    for name, extra in (("fixed", ["--capacity-log2", "24"]), ("growing", ["--initial-capacity-log2", "14"])):
        out = tmp_path / f"{name}.csv"
        p = subprocess.run([sys.executable, os.path.join(REPO, "train.py"), *common, *extra, "--log", str(out)],
                           capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
        assert p.returncode == 0, p.stderr[-3000:]
        runs[name] = (p.stdout + p.stderr, [r[:3] + r[4:7] for r in list(csv.reader(open(out)))[1:]])
    assert runs["growing"][0].count("table check passed") == 2 and runs["growing"][0].count("table grew") >= 4
    assert runs["growing"][1] == runs["fixed"][1] and len(runs["fixed"][1]) >= 2
