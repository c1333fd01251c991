"""ctypes/numpy front-end of the CPU ORACLE (oracle/q2048_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; the product package never imports this module.  See q2048_oracle.h for
what the oracle restates and how it is pinned to the reference.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

MAXCELLS = 32
NO_ACTION = -1
STREAM_STEP, STREAM_RESET, STREAM_OVER = 0, 1, 2
ENV_DQN, ENV_RESET_SHAPING, ENV_NEW_VISITS = 1, 2, 4     # env_flags of rollout()
ST_STEPS, ST_EPISODES, ST_VALID, ST_SCORE, ST_INSERTS, ST_DROPS, ST_EXPLORE = range(7)
ST_HIST0, ST_NI = 8, 32
SF_RETURN, SF_RETURN_SQ, SF_REWARD, SF_NF = 0, 1, 2, 4

# numpy mirror of orc_env_t (168 bytes)
ENV_DTYPE = np.dtype(
    [
        ("board", np.uint8, (MAXCELLS,)),
        ("n", np.int32),
        ("previous_max_log2", np.int32),
        ("score", np.int64),
        ("move_score", np.int64),
        ("consecutive_action", np.int32),
        ("pad0", np.int32),
        ("consecutive_count", np.int64),
        ("last_consecutive_penalty", np.float64),
        ("episode_return", np.float64),
        ("episode", np.uint32),
        ("pad1", np.uint32),
        ("visit_key", np.uint8, (MAXCELLS,)),      # orc_visit_t: the env's visit row under a closed key set
        ("visit_q", np.float64, (4,)),
        ("visit_valid", np.int32),
        ("visit_pad", np.int32),
    ],
    align=True,
)


def build(force: bool = False) -> str:
    """Compile liboracle.so with the committed Makefile (gcc, seconds)."""
    src = os.path.join(_HERE, "q2048_oracle.c")
    hdr = os.path.join(_HERE, "q2048_oracle.h")
    stale = (
        force
        or not os.path.exists(_LIB_PATH)
        or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))
    )
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    u8p, u32p, i64p, f64p = (C.POINTER(C.c_uint8), C.POINTER(C.c_uint32),
                             C.POINTER(C.c_int64), C.POINTER(C.c_double))
    vp = C.c_void_p
    sig = {
        "orc_philox4x32_10": (None, [u32p, u32p, u32p]),
        "orc_draws": (None, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, u32p]),
        "orc_draw_uniform": (C.c_double, [C.c_uint32]),
        "orc_draw_action": (C.c_int, [C.c_uint32]),
        "orc_draw_index": (C.c_int, [C.c_uint32, C.c_int]),
        "orc_draw_is_four": (C.c_int, [C.c_uint32]),
        "orc_move_left_line": (C.c_int, [u8p, C.c_int, i64p]),
        "orc_rotate_ccw": (None, [u8p, C.c_int]),
        "orc_move": (C.c_int, [u8p, C.c_int, C.c_int, i64p]),
        "orc_count_empty": (C.c_int, [u8p, C.c_int]),
        "orc_add_number": (C.c_int, [u8p, C.c_int, C.c_uint32, C.c_uint32]),
        "orc_add_number_at": (None, [u8p, C.c_int, C.c_int, C.c_int]),
        "orc_is_game_over": (C.c_int, [u8p, C.c_int]),
        "orc_max_log2": (C.c_int, [u8p, C.c_int]),
        "orc_update_and_normalize": (C.c_double, [C.c_double]),
        "orc_calculate_reward": (C.c_double, [vp, C.c_int64, C.c_int, C.c_int, C.c_int]),
        "orc_env_init": (None, [vp, C.c_int, u32p]),
        "orc_env_reset": (None, [vp, u32p]),
        "orc_env_step": (C.c_int, [vp, C.c_int, C.c_uint32, C.c_uint32, f64p,
                                   C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "orc_agent_new": (vp, [C.c_double, C.c_int, C.c_double, C.c_double, C.c_double,
                               C.c_double, C.c_int]),
        "orc_agent_free": (None, [vp]),
        "orc_agent_reserve": (None, [vp, C.c_int64]),
        "orc_agent_choose": (C.c_int, [vp, u8p, C.c_uint32, C.c_uint32, C.POINTER(C.c_int)]),
        "orc_agent_update": (None, [vp, u8p, C.c_int, C.c_double, u8p, C.c_int]),
        "orc_agent_decay": (None, [vp, C.c_double]),
        "orc_agent_q": (C.c_int, [vp, u8p, f64p]),
        "orc_agent_size": (C.c_int64, [vp]),
        "orc_agent_set_storage_f32": (None, [vp, C.c_int]),
        "orc_agent_set_frozen": (None, [vp, C.c_int]),
        "orc_agent_drops": (C.c_int64, [vp]),
        "orc_agent_dump": (C.c_int64, [vp, u8p, f64p, C.c_int64]),
        "orc_envs_init": (None, [vp, C.c_int64, C.c_int, C.c_uint64, C.c_uint64]),
        "orc_rollout": (None, [vp, C.c_int64, vp, C.c_int64, C.c_uint64, C.c_uint64,
                               C.c_uint32, u8p, i64p, f64p, u8p, f64p, u8p]),
        "orc_rollout_sync": (None, [vp, C.c_int64, vp, C.c_int64, C.c_uint64, C.c_uint64, C.c_uint32,
                                    i64p, f64p]),
        "orc_rollout_mt": (None, [vp, C.c_int64, C.POINTER(vp), C.c_int, C.c_int64,
                                  C.c_uint64, C.c_uint64, C.c_uint32, i64p, f64p]),
        "orc_rt_new": (vp, []),
        "orc_rt_free": (None, [vp]),
        "orc_rt_q": (None, [vp, u8p, C.POINTER(C.c_float)]),
        "orc_rt_choose": (C.c_int, [vp, u8p, C.c_double, C.c_uint32, C.c_uint32, C.POINTER(C.c_int)]),
        "orc_rt_update": (None, [vp, u8p, C.c_int, C.c_float, u8p, C.c_int, C.c_double, C.c_double]),
        "orc_rt_rollout": (None, [vp, C.c_int64, vp, C.c_int64, C.c_double, C.c_double, C.c_double,
                                  C.c_uint64, C.c_uint64, C.c_uint32, i64p, f64p]),
        "orc_env_step_dqn": (C.c_int, [vp, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, f64p,
                                       C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "orc_env_reset_shaping": (None, [vp]),
        "orc_rollout_ex": (None, [vp, C.c_int64, vp, C.c_int64, C.c_uint64, C.c_uint64,
                                  C.c_uint32, u8p, i64p, f64p, u8p, f64p, u8p, C.c_int]),
        "orc_sizeof_env": (C.c_int, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    assert L.orc_sizeof_env() == ENV_DTYPE.itemsize, (L.orc_sizeof_env(), ENV_DTYPE.itemsize)
    _lib = L
    return L


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _u32(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def _ptr(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


# ---- RNG ----------------------------------------------------------------------------
def philox(ctr, key) -> np.ndarray:
    c = np.asarray(ctr, dtype=np.uint32).copy()
    k = np.asarray(key, dtype=np.uint32).copy()
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32_10(_u32(c), _u32(k), _u32(out))
    return out


def draws(seed: int, env_id: int, ctr: int, stream: int = STREAM_STEP) -> np.ndarray:
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_draws(seed, env_id, ctr & 0xFFFFFFFF, stream, _u32(out))
    return out


def draw_uniform(x: int) -> float:
    return lib().orc_draw_uniform(int(x))


def draw_action(x: int) -> int:
    return lib().orc_draw_action(int(x))


def draw_index(x: int, n: int) -> int:
    return lib().orc_draw_index(int(x), int(n))


def draw_is_four(x: int) -> bool:
    return bool(lib().orc_draw_is_four(int(x)))


# ---- game core ----------------------------------------------------------------------
def move_left_line(line) -> tuple[np.ndarray, int, bool]:
    a = np.asarray(line, dtype=np.uint8).copy()
    s = C.c_int64(0)
    moved = lib().orc_move_left_line(_u8(a), a.size, C.byref(s))
    return a, s.value, bool(moved)


def move(board, action: int, n: int = 4) -> tuple[np.ndarray, int, bool]:
    """board: uint8[n*n] log2.  Returns (board', score, moved); no spawn."""
    a = np.asarray(board, dtype=np.uint8).reshape(-1).copy()
    s = C.c_int64(0)
    moved = lib().orc_move(_u8(a), n, int(action), C.byref(s))
    if moved < 0:
        raise ValueError(f"action {action} outside 0..3")
    return a, s.value, bool(moved)


def add_number_at(board, k: int, is_four: bool, n: int = 4) -> np.ndarray:
    a = np.asarray(board, dtype=np.uint8).reshape(-1).copy()
    lib().orc_add_number_at(_u8(a), n, int(k), int(is_four))
    return a


def add_number(board, draw_pos: int, draw_val: int, n: int = 4) -> np.ndarray:
    a = np.asarray(board, dtype=np.uint8).reshape(-1).copy()
    lib().orc_add_number(_u8(a), n, int(draw_pos), int(draw_val))
    return a


def is_game_over(board, n: int = 4) -> bool:
    a = np.ascontiguousarray(np.asarray(board, dtype=np.uint8).reshape(-1))
    return bool(lib().orc_is_game_over(_u8(a), n))


def update_and_normalize(r: float) -> float:
    return lib().orc_update_and_normalize(float(r))


# ---- env ----------------------------------------------------------------------------
class Env:
    """One reference-shaped env (Game2048_env) driven by explicit draws."""

    def __init__(self, n: int = 4, draws4=(0, 0, 0, 0)):
        self.n = n
        self.rec = np.zeros(1, dtype=ENV_DTYPE)
        d = np.asarray(draws4, dtype=np.uint32).copy()
        lib().orc_env_init(self.rec.ctypes.data, n, _u32(d))

    @property
    def board(self) -> np.ndarray:
        return self.rec["board"][0, : self.n * self.n].copy()

    def set_board(self, board):
        b = np.zeros(MAXCELLS, dtype=np.uint8)
        b[: self.n * self.n] = np.asarray(board, dtype=np.uint8).reshape(-1)
        self.rec["board"][0] = b

    def reset(self, draws4):
        d = np.asarray(draws4, dtype=np.uint32).copy()
        lib().orc_env_reset(self.rec.ctypes.data, _u32(d))
        return self.board

    def step(self, action: int, draw_pos: int = 0, draw_val: int = 0):
        r, d, m = C.c_double(0), C.c_int(0), C.c_int(0)
        v = lib().orc_env_step(self.rec.ctypes.data, int(action), int(draw_pos), int(draw_val),
                               C.byref(r), C.byref(d), C.byref(m))
        if v < 0:
            raise ValueError(f"action {action} outside 0..3")
        return self.board, r.value, bool(d.value), m.value, bool(v)

    def step_dqn(self, action: int, draw_pos: int = 0, draw_val: int = 0, over_pos: int = 0,
                 over_val: int = 0):
        """Game2048_nopenalty_env.step (+ the caller's board write-back): see orc_env_step_dqn."""
        r, d, m = C.c_double(0), C.c_int(0), C.c_int(0)
        v = lib().orc_env_step_dqn(self.rec.ctypes.data, int(action), int(draw_pos), int(draw_val),
                                   int(over_pos), int(over_val), C.byref(r), C.byref(d), C.byref(m))
        if v < 0:
            raise ValueError(f"action {action} outside 0..3")
        return self.board, r.value, bool(d.value), m.value, bool(v)

    def calculate_reward(self, score, valid, game_over, max_log2) -> float:
        return lib().orc_calculate_reward(self.rec.ctypes.data, int(score), int(valid),
                                          int(game_over), int(max_log2))


# ---- agent --------------------------------------------------------------------------
class Agent:
    def __init__(self, total_epochs, action_space=4, learning_rate=0.1, discount_factor=0.9,
                 exploration_rate=1.0, exploration_min=0.01, n: int = 4, storage_f32: bool = False):
        """storage_f32: rows hold float32 values (the device's storage type; a documented deviation
        from the reference's float64 dict, see orc_agent_t.storage_f32)."""
        self.n = n
        self._h = lib().orc_agent_new(float(total_epochs), action_space, learning_rate,
                                      discount_factor, exploration_rate, exploration_min, n)
        if storage_f32:
            lib().orc_agent_set_storage_f32(self._h, 1)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:  # _lib is gone at interpreter exit
            _lib.orc_agent_free(self._h)
            self._h = None

    class _View(C.Structure):
        _fields_ = [(k, C.c_double) for k in (
            "lr", "gamma", "epsilon", "epsilon_min", "total_epochs", "first_decay_limit",
            "second_decay_limit", "third_decay_limit", "slow_decay_1", "fast_decay",
            "slow_decay_2")]

    def _view(self):
        return C.cast(self._h, C.POINTER(Agent._View)).contents

    @property
    def epsilon(self) -> float:
        return self._view().epsilon

    @epsilon.setter
    def epsilon(self, v: float):
        self._view().epsilon = float(v)

    def reserve(self, rows: int):
        lib().orc_agent_reserve(self._h, int(rows))

    def freeze(self, on: bool = True):
        """Closed key set (orc_agent_t.frozen; NOT the reference -- the build's policy for a table that cannot
        grow, Q2048_FLAG_NO_NEW_ROWS): absent states read as zeros and are not created, their updates drop."""
        lib().orc_agent_set_frozen(self._h, int(bool(on)))

    @property
    def drops(self) -> int:
        return int(lib().orc_agent_drops(self._h))

    def _key(self, board):
        b = np.zeros(MAXCELLS, dtype=np.uint8)
        b[: self.n * self.n] = np.asarray(board, dtype=np.uint8).reshape(-1)
        return b

    def choose_action(self, board, draw_eps: int, draw_act: int) -> int:
        return lib().orc_agent_choose(self._h, _u8(self._key(board)), int(draw_eps),
                                      int(draw_act), None)

    def update_q_value(self, s, action, reward, s2, done):
        lib().orc_agent_update(self._h, _u8(self._key(s)), int(action), float(reward),
                               _u8(self._key(s2)), int(bool(done)))

    def decay_exploration(self, epoch):
        lib().orc_agent_decay(self._h, float(epoch))

    def q(self, board) -> np.ndarray:
        out = np.zeros(4, dtype=np.float64)
        lib().orc_agent_q(self._h, _u8(self._key(board)), _ptr(out, C.c_double))
        return out

    def q_many(self, boards) -> np.ndarray:
        boards = np.asarray(boards, dtype=np.uint8).reshape(len(boards), -1)
        return np.stack([self.q(b) for b in boards]) if len(boards) else np.zeros((0, 4))

    def __len__(self) -> int:
        return int(lib().orc_agent_size(self._h))

    def dump(self):
        n = len(self)
        keys = np.zeros((n, MAXCELLS), dtype=np.uint8)
        vals = np.zeros((n, 4), dtype=np.float64)
        w = lib().orc_agent_dump(self._h, _u8(keys), _ptr(vals, C.c_double), n)
        return keys[:w, : self.n * self.n], vals[:w]


# ---- batched driver -----------------------------------------------------------------
def envs_init(B: int, n: int = 4, seed: int = 0, env_id0: int = 0) -> np.ndarray:
    envs = np.zeros(B, dtype=ENV_DTYPE)
    lib().orc_envs_init(envs.ctypes.data, B, n, seed, env_id0)
    return envs


def rollout(envs: np.ndarray, agent: Agent | None, steps: int, seed: int = 0,
            env_id0: int = 0, ctr0: int = 0, actions: np.ndarray | None = None,
            record: bool = False, env_flags: int = 0):
    """Runs `steps` lockstep steps in place.  Returns (stats_i, stats_f[, acts, rew, done]).
    env_flags: ENV_DQN = the DQN path's env (calculate_reward2, done = game_over),
    ENV_RESET_SHAPING = resets also restore the shaping state, ENV_NEW_VISITS = (closed key set) the envs' visit
    rows end when the call begins, as on a device launch without a row cache."""
    B = len(envs)
    si = np.zeros(ST_NI, dtype=np.int64)
    sf = np.zeros(SF_NF, dtype=np.float64)
    acts = rew = dn = None
    if record:
        acts = np.zeros((steps, B), dtype=np.uint8)
        rew = np.zeros((steps, B), dtype=np.float64)
        dn = np.zeros((steps, B), dtype=np.uint8)
    if actions is not None:
        actions = np.ascontiguousarray(actions, dtype=np.uint8).reshape(steps, B)
    lib().orc_rollout_ex(envs.ctypes.data, B, agent._h if agent is not None else None, steps,
                         seed, env_id0, ctr0 & 0xFFFFFFFF, _ptr(actions, C.c_uint8),
                         _ptr(si, C.c_int64), _ptr(sf, C.c_double), _ptr(acts, C.c_uint8),
                         _ptr(rew, C.c_double), _ptr(dn, C.c_uint8), int(env_flags))
    if record:
        return si, sf, acts, rew, dn
    return si, sf


def rollout_sync(envs: np.ndarray, agent: Agent, steps: int, seed: int = 0, env_id0: int = 0,
                 ctr0: int = 0):
    """The deterministic two-phase batched semantic (see orc_rollout_sync)."""
    si = np.zeros(ST_NI, dtype=np.int64)
    sf = np.zeros(SF_NF, dtype=np.float64)
    lib().orc_rollout_sync(envs.ctypes.data, len(envs), agent._h, steps, seed, env_id0,
                           ctr0 & 0xFFFFFFFF, _ptr(si, C.c_int64), _ptr(sf, C.c_double))
    return si, sf


def rollout_mt(envs: np.ndarray, agents: "list[Agent] | None", steps: int, seed: int = 0,
               env_id0: int = 0, ctr0: int = 0, threads: int = 0):
    """T threads over contiguous env ranges: one private learner each (`agents`), or, with
    agents=None, `threads` threads of uniformly random play (no learner)."""
    T = len(agents) if agents is not None else int(threads)
    hs = (C.c_void_p * T)(*[a._h for a in agents]) if agents is not None else None
    si = np.zeros(ST_NI, dtype=np.int64)
    sf = np.zeros(SF_NF, dtype=np.float64)
    lib().orc_rollout_mt(envs.ctypes.data, len(envs), hs, T, steps, seed, env_id0,
                         ctr0 & 0xFFFFFFFF, _ptr(si, C.c_int64), _ptr(sf, C.c_double))
    return si, sf


class RowTupleAgent:
    """The row-tuple linear Q learner of BASELINE configs[1] (not in the reference)."""

    def __init__(self, learning_rate=0.1, discount_factor=0.9, exploration_rate=1.0):
        self.lr, self.gamma, self.epsilon = learning_rate, discount_factor, exploration_rate
        self._w = lib().orc_rt_new()

    def __del__(self):
        if getattr(self, "_w", None) and _lib is not None:
            _lib.orc_rt_free(self._w)
            self._w = None

    def _b(self, board):
        b = np.zeros(MAXCELLS, dtype=np.uint8)
        b[:16] = np.asarray(board, dtype=np.uint8).reshape(-1)
        return b

    def q(self, board) -> np.ndarray:
        out = np.zeros(4, dtype=np.float32)
        lib().orc_rt_q(self._w, _u8(self._b(board)), _ptr(out, C.c_float))
        return out

    def weights(self) -> np.ndarray:
        buf = (C.c_float * (4 * 65536 * 4)).from_address(self._w)
        return np.frombuffer(buf, dtype=np.float32).reshape(4, 65536, 4).copy()

    def choose_action(self, board, draw_eps, draw_act) -> int:
        return lib().orc_rt_choose(self._w, _u8(self._b(board)), self.epsilon, int(draw_eps),
                                   int(draw_act), None)

    def update_q_value(self, s, action, reward, s2, done):
        lib().orc_rt_update(self._w, _u8(self._b(s)), int(action), float(reward), _u8(self._b(s2)),
                            int(bool(done)), self.lr, self.gamma)

    def rollout(self, envs: np.ndarray, steps: int, seed=0, env_id0=0, ctr0=0):
        si = np.zeros(ST_NI, dtype=np.int64)
        sf = np.zeros(SF_NF, dtype=np.float64)
        lib().orc_rt_rollout(envs.ctypes.data, len(envs), self._w, steps, self.epsilon, self.lr,
                             self.gamma, seed, env_id0, ctr0 & 0xFFFFFFFF, _ptr(si, C.c_int64),
                             _ptr(sf, C.c_double))
        return si, sf


# ---- reference <-> oracle board conversion -------------------------------------------
def to_log2(raw) -> np.ndarray:
    """np.int64 raw tile values (reference) -> uint8 log2 (0 stays 0)."""
    raw = np.asarray(raw, dtype=np.int64)
    out = np.zeros(raw.shape, dtype=np.uint8)
    nz = raw > 0
    out[nz] = np.round(np.log2(raw[nz])).astype(np.uint8)
    assert np.all((1 << out[nz].astype(np.int64)) == raw[nz])
    return out


def to_raw(log2b) -> np.ndarray:
    b = np.asarray(log2b, dtype=np.int64)
    return np.where(b > 0, 1 << b, 0).astype(np.int64)
