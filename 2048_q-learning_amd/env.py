"""Batched 2048 environment on MI355X: the host-side mirror of
QLearningBase/environment/Game2048_env.py (class Game2048_env, :78-205).

`BatchedGame2048Env` keeps the reference's method names (`reset`, `step`, `action_space.n`,
`score`) over B boards resident in HBM; `Game2048_env` is the one-env adapter that returns
the reference's Python types so that the loop body of Agent/main.py:91-101 runs unchanged.
All computation happens behind the C ABI of include/q2048.h -- the HIP kernels for device "cuda[:i]",
their CPU twin (libq2048_host.so: the same ABI from the same per-lane arithmetic) for the explicit
device "cpu"; torch only owns the memory and the stream.  The device is the caller's choice and is
never substituted."""
from __future__ import annotations

import numpy as np
import torch

from . import _native as N


class _Discrete:
    """Stand-in for gymnasium.spaces.Discrete: the reference only reads `.n`
    (Game2048_env.py:89, Agent/main.py:68)."""

    def __init__(self, n: int):
        self.n = n


def _ptr(t: torch.Tensor | None):
    return None if t is None else t.data_ptr()


def _stream(device: torch.device):
    """The caller's current stream as the ABI's `stream` argument (host library: ignored, NULL)."""
    return torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else None


def _sync(device: torch.device) -> None:
    """Waits for the caller's stream (host library: every call is complete when it returns)."""
    if device.type == "cuda":
        torch.cuda.current_stream(device).synchronize()


def _host_zeros(shape, dtype, device: torch.device) -> torch.Tensor:
    """Host memory the kernels can address directly: pinned (mapped into the device's address space) for a
    GPU, plain for the host library."""
    t = torch.zeros(shape, dtype=dtype)
    return t.pin_memory() if device.type == "cuda" else t


def _require_gpu(device) -> torch.device:
    """The device a caller named, checked: "cuda[:i]" needs a visible HIP device and the HIP library, "cpu"
    (the explicit CPU twin) the host library.  Whichever is missing raises -- nothing is substituted.
    (The name is historical: until round 5 only GPUs passed.)"""
    device = torch.device(device)
    if device.type == "cpu":
        N.host_lib()
        return device
    if device.type != "cuda":
        raise RuntimeError(f"2048_q-learning_amd runs on 'cuda[:i]' (MI355X) or, by explicit request, 'cpu'; "
                           f"not on {device}")
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible: the HIP extension cannot run (device 'cpu' must be asked "
                           "for by name)")
    N.lib()
    return device


class BatchedGame2048Env:
    """B independent boards (4x4, or 5x5 with board_size=5), one per GPU lane.

    boards  torch.uint8 [B, n*n] log2 tiles, row-major (0 empty, k = tile 2^k)
    aux     torch.uint8 [B, 16]  q2048_aux records (score, return, previous_max, streak, episode)

    Lane i is global env `env_id0 + i`; its random draws depend only on (seed, global id,
    step counter), never on B or on how a batch is sharded over GPUs.

    `step` ping-pongs two board buffers: it reads the current one and writes the other, which
    becomes `env.boards`.  The tensor that was `env.boards` before the call (what `reset()` or the
    previous `step()` returned: the loop's `state`) therefore still holds the pre-step boards, intact
    until the NEXT step -- `agent.update_q_value(state, a, r, next_state, done)` needs no
    `state.clone()` (Agent/main.py:92-100 in batched form).  Rollout entry points work in place."""

    def __init__(self, num_envs: int, board_size: int = 4, device="cuda", seed: int = 0,
                 env_id0: int = 0, profile: str = "shaped", reset_shaping_state: bool = False,
                 host_visible: bool = False, copy_outputs: bool = False):
        """profile  "shaped" = Game2048_env of QLearningBase (the hot path's env);
                    "nopenalty" = the DQN path's Game2048_env,
                    Deep_QLearning/environment/Game2048_nopenalty_env.py: reward =
                    calculate_reward2 (-10 invalid-not-over, else the merge score), done =
                    game_over, with that file's is_game_over quirks (include/q2048.h,
                    Q2048_FLAG_ENV_DQN).
        reset_shaping_state  opt-in fix of a reference bug (Game2048_env.py:187-191 keeps
                    previous_max and the consecutive-action streak across reset()): resets also
                    restore them to their constructor values.  Default: the reference's behaviour.
        host_visible  keep boards / aux / outputs in pinned host memory that the kernels address
                    directly (the one-env adapter: a step is one launch and one synchronisation,
                    no copies).  For a handful of envs only: every access crosses PCIe.
        copy_outputs  `step` returns fresh reward / done / max_tile tensors instead of views of the
                    env's own output buffers.  Default off (since round 3): the views are what keeps
                    every torch kernel out of the batched loop, and the loop of Agent/main.py:92-100
                    consumes a step's outputs before the next step; a caller that KEEPS them across
                    steps (a list of rewards, a comparison with the previous `done`) must turn this on
                    or clone.  `boards` is always one of two ping-pong buffers: the tensor returned by
                    step t is overwritten by step t + 2."""
        self.device = _require_gpu(device)
        self._L = N.lib_for(self.device)
        self.copy_outputs = bool(copy_outputs)
        if board_size not in (4, 5):
            raise NotImplementedError("board_size must be 4 (the reference) or 5")
        if num_envs <= 0:
            raise ValueError("num_envs must be positive")
        if profile not in ("shaped", "nopenalty"):
            raise ValueError("profile must be 'shaped' or 'nopenalty'")
        self.profile, self.reset_shaping_state = profile, bool(reset_shaping_state)
        self.env_flags = ((N.FLAG_ENV_DQN if profile == "nopenalty" else 0) |
                          (N.FLAG_RESET_SHAPING if reset_shaping_state else 0))
        self.num_envs, self.board_size = int(num_envs), int(board_size)
        self.cells = self.board_size * self.board_size
        self.seed, self.env_id0 = int(seed), int(env_id0)
        self.ctr = 0  # global step counter = counter word of the step draws
        self.action_space = _Discrete(4)                                 # Game2048_env.py:89
        B = self.num_envs
        self.host_visible = bool(host_visible)

        def alloc(shape, dtype):
            if self.host_visible:                       # pinned + mapped: same address on the device
                return _host_zeros(shape, dtype, self.device)
            return torch.zeros(shape, dtype=dtype, device=self.device)

        # one buffer for the host-visible one-env adapter (its step is in place), two otherwise
        self._bufs = [alloc((B, self.cells), torch.uint8) for _ in range(1 if self.host_visible else 2)]
        self._cur = 0
        self.aux = alloc((B, 16), torch.uint8)
        self._reward = alloc(B, torch.float32)
        self._done = alloc(B, torch.uint8)
        self._max = alloc(B, torch.uint8)
        self._max_tile = alloc(B, torch.int32)
        self.status = alloc(1, torch.int32)
        N.check(self._L.q2048_env_init(_ptr(self.boards), _ptr(self.aux), B, self.board_size,
                                       self.seed, self.env_id0, _stream(self.device)), "env_init")
        if self.host_visible:
            _sync(self.device)

    @property
    def boards(self) -> torch.Tensor:
        """The current boards, uint8 [B, n*n] (the buffer the last step wrote)."""
        return self._bufs[self._cur]

    # -- reference surface ---------------------------------------------------------------
    def reset(self, mask: torch.Tensor | None = None) -> torch.Tensor:
        """Game2048_env.reset (:187-191) for the lanes where mask != 0 (all when None)."""
        if mask is not None:
            mask = self._as_u8(mask, "mask")
        N.check(self._L.q2048_env_reset_ex(_ptr(self.boards), _ptr(self.aux), _ptr(mask),
                                           self.num_envs, self.board_size, self.seed, self.env_id0,
                                           self.env_flags, _stream(self.device)), "env_reset")
        if self.host_visible:
            _sync(self.device)
        return self.boards

    def step(self, actions: torch.Tensor):
        """Game2048_env.step (:97-129): returns (boards, reward[B] f32, done[B] bool,
        max_tile[B] int32 raw tile value, the reference's `info`).  The outputs are views of buffers
        the next step overwrites; `boards` is the other board buffer (see the class docstring)."""
        actions = self._as_u8(actions, "actions")
        src = self._bufs[self._cur]
        nxt = (self._cur + 1) % len(self._bufs)
        N.check(self._L.q2048_env_step_to(
            _ptr(src), _ptr(self._bufs[nxt]), _ptr(self.aux), _ptr(actions), self.num_envs, self.board_size,
            self.seed, self.env_id0, self.ctr & 0xFFFFFFFF, self.env_flags, _ptr(self._reward),
            _ptr(self._done), _ptr(self._max), _ptr(self._max_tile), _ptr(self.status),
            _stream(self.device)), "env_step")
        self._cur = nxt
        self.ctr += 1
        if self.host_visible:                           # host tensors: the launch has to finish first
            _sync(self.device)
        if self.copy_outputs:
            return self.boards, self._reward.clone(), self._done.view(torch.bool).clone(), self._max_tile.clone()
        return self.boards, self._reward, self._done.view(torch.bool), self._max_tile

    @property
    def score(self) -> torch.Tensor:
        """env.score (:84) per lane."""
        return self.aux.view(torch.int32)[:, 0]

    @property
    def max_log2(self) -> torch.Tensor:
        return self._max

    # -- bridges to the reference's DQN front-end ------------------------
    def legal_moves(self) -> torch.Tensor:
        """uint8 [B]: bit a set iff action a would change the board -- the trial-move loop of
        Deep_QLearning/main_dir/mainDQL_CNN_step2.py:168-174."""
        mask = torch.empty(self.num_envs, dtype=torch.uint8, device=self.device)
        N.check(self._L.q2048_legal_moves(_ptr(self.boards), self.num_envs, self.board_size,
                                          _ptr(mask), _stream(self.device)), "legal_moves")
        return mask

    def encode_onehot(self, dtype: torch.dtype = torch.float32) -> torch.Tensor:
        """[B, 16, 4, 4] one-hot of the log2 tiles (Dqn8TestNOPERCNN.py:271-277); 4x4 only."""
        if self.board_size != 4:
            raise ValueError("the one-hot encoder is defined for 4x4 boards")
        if dtype not in (torch.float32, torch.bfloat16):
            raise TypeError("dtype must be float32 or bfloat16")
        out = torch.empty((self.num_envs, 16, 4, 4), dtype=dtype, device=self.device)
        N.check(self._L.q2048_encode_onehot(_ptr(self.boards), self.num_envs,
                                            0 if dtype == torch.float32 else 1, _ptr(out),
                                            _stream(self.device)), "encode_onehot")
        return out

    # -- checkpoint ---------------------------------------------------------------------------
    def state_dict(self) -> dict:
        """Everything needed to continue this batch bit-exactly (host tensors)."""
        return {"boards": self.boards.to("cpu", copy=True), "aux": self.aux.to("cpu", copy=True), "ctr": self.ctr,
                "seed": self.seed, "env_id0": self.env_id0, "board_size": self.board_size,
                "profile": self.profile, "reset_shaping_state": self.reset_shaping_state}

    def load_state_dict(self, sd: dict) -> None:
        if (sd["board_size"], tuple(sd["boards"].shape)) != (self.board_size, tuple(self.boards.shape)):
            raise ValueError("checkpoint was taken with another batch or board size")
        self.boards.copy_(sd["boards"])
        self.aux.copy_(sd["aux"])
        self.ctr, self.seed, self.env_id0 = int(sd["ctr"]), int(sd["seed"]), int(sd["env_id0"])
        if (sd.get("profile", "shaped"), sd.get("reset_shaping_state", False)) != (
                self.profile, self.reset_shaping_state):
            raise ValueError("checkpoint was taken with another env profile")

    # -- helpers --------------------------------------------------------------------------
    def aux_fields(self) -> dict:
        """Host copy of the aux records as numpy fields (tests / logging)."""
        a = self.aux.cpu().numpy().view(AUX_DTYPE).reshape(-1)
        return {k: a[k].copy() for k in AUX_DTYPE.names}

    def check_status(self) -> int:
        """Synchronising read of the device status word; raises on a rejected action."""
        if self.host_visible:
            _sync(self.device)
        s = int(self.status.item())
        if s & N.STATUS_BAD_ACTION:
            self.status.zero_()
            raise ValueError("an action outside 0..3 was passed to step() (lane left untouched)")
        return s

    def _as_u8(self, t, name: str) -> torch.Tensor:
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(t)
        if t.dtype == torch.bool:
            t = t.view(torch.uint8)                     # same bytes: no copy, no kernel
        if t.dtype != torch.uint8:
            if t.numel() and (int(t.min()) < 0 or int(t.max()) > 255):
                raise ValueError(f"{name} out of range")
            t = t.to(torch.uint8)
        if self.host_visible and self.device.type == "cuda":
            t = t.cpu().pin_memory().contiguous()
        else:
            t = t.to(self.device).contiguous()
        if t.shape != (self.num_envs,):
            raise ValueError(f"{name} must have shape ({self.num_envs},), got {tuple(t.shape)}")
        return t


AUX_DTYPE = np.dtype([("score", "<i4"), ("ep_return", "<f4"), ("prev_max", "u1"),
                      ("cons_action", "u1"), ("cons_count", "<u2"), ("episode", "<u4")])


_RAW_OF_LOG2 = np.array([0] + [1 << k for k in range(1, 63)] + [0] * 193, dtype=np.int64)   # (one gather; called per step by the adapters)


def boards_to_raw(boards_log2) -> np.ndarray:
    """uint8 log2 boards [..., n*n] -> np.int64 raw tile values [..., n, n] (reference layout)."""
    b = np.asarray(boards_log2)
    if b.dtype != np.uint8:
        b = b.astype(np.int64)
        if b.size and (int(b.min()) < 0 or int(b.max()) > 62):
            raise ValueError("log2 tiles lie in 0..62")
    n = 4 if b.shape[-1] == 16 else 5 if b.shape[-1] == 25 else int(round(b.shape[-1] ** 0.5))
    return _RAW_OF_LOG2[b].reshape(b.shape[:-1] + (n, n))


def raw_to_boards(raw) -> np.ndarray:
    """Reference boards (raw tile values, any nesting of n x n) -> uint8 log2 [..., n*n]."""
    r = np.asarray(raw, dtype=np.int64)
    if r.ndim >= 2 and r.shape[-1] == r.shape[-2] and r.shape[-1] in (4, 5):
        r = r.reshape(r.shape[:-2] + (r.shape[-1] * r.shape[-1],))
    out = np.zeros(r.shape, dtype=np.uint8)
    nz = r > 0
    lg = np.zeros(r.shape, dtype=np.int64)
    lg[nz] = np.round(np.log2(r[nz])).astype(np.int64)
    if np.any(np.left_shift(1, lg[nz]) != r[nz]):
        raise ValueError("board holds a value that is not a power of two")
    out[nz] = lg[nz]
    return out


class _Game:
    """`env.game.board` of the reference (Agent/main.py:85-86)."""

    def __init__(self, env: "Game2048_env"):
        self._env = env

    @property
    def board(self) -> np.ndarray:
        return boards_to_raw(self._env._b.boards.numpy())[0]        # pinned host memory, in sync


_LOG2_OF = {0: 0, **{1 << k: k for k in range(1, 32)}}


def state_to_log2(state, out: np.ndarray) -> None:
    """One reference state (4 x 4 raw tile values: array or tuple of tuples) -> 16 log2 bytes in
    `out`; the scalar twin of raw_to_boards for the one-env adapters."""
    k = 0
    try:
        for row in state:
            for v in row:
                out[k] = _LOG2_OF[int(v)]
                k += 1
    except KeyError:
        raise ValueError("board holds a value that is not a power of two") from None
    if k != 16:
        raise ValueError("a state is a 4 x 4 board")


class _Staging:
    """A ring of pinned host memory that the kernels address directly (pinned host memory is mapped
    into the device's address space at the same address): the one-env adapters write their inputs
    with numpy, launch, and read the outputs after one stream synchronisation -- no copy calls.
    `take(n)` hands out the next n bytes; a region is not handed out again before the stream has
    been synchronised, so a kernel still in flight never sees its inputs overwritten."""

    def __init__(self, device: torch.device, nbytes: int = 1 << 14):
        self.device = device
        self.host = _host_zeros(nbytes, torch.uint8, device)
        self.np = self.host.numpy()
        self.ptr = self.host.data_ptr()
        self.pos = 0
        self.pending = []                       # results not yet read back (see agent._LazyRow)

    def take(self, nbytes: int) -> int:
        nbytes = (nbytes + 15) & ~15
        if self.pos + nbytes > self.np.size:
            self.sync()
            self.pos = 0
        off, self.pos = self.pos, self.pos + nbytes
        return off

    def sync(self) -> None:
        _sync(self.device)
        pending, self.pending = self.pending, []
        for ref in pending:
            obj = ref()
            if obj is not None:
                obj._read()


class Game2048_env:
    """One env with the reference's exact surface and Python types
    (Game2048_env.py:78-205): reset() -> int64[4,4], step(a) -> (board, float, bool, int).
    Board, aux and the step's outputs live in pinned host memory (`host_visible`), so a step is
    one kernel launch and one stream synchronisation."""

    def __init__(self, device="cuda", seed: int = 0, env_id: int = 0, profile: str = "shaped",
                 reset_shaping_state: bool = False, play_first_board: bool = False):
        """play_first_board  the reset() that the loop makes before its FIRST episode (Agent/main.py:81) returns
                    the constructor's game instead of drawing another one.  The reference throws that first game
                    away unplayed (a second pair of spawns, statistically the same board); with this flag the
                    one-env loop plays exactly what lane 0 of the batched rollout plays -- and what the golden
                    transcript G6 recorded from the reference -- draw for draw.  Default off: the reference's
                    call sequence."""
        self._keep_first = bool(play_first_board)
        self._b = BatchedGame2048Env(1, 4, device, seed, env_id, profile=profile,
                                     reset_shaping_state=reset_shaping_state, host_visible=True)
        self.action_space = self._b.action_space
        self.game = _Game(self)
        b = self._b
        self._acts = _host_zeros(4, torch.uint8, b.device)
        self._acts.copy_(torch.arange(4, dtype=torch.uint8))
        self._board_np = b.boards.numpy().reshape(16)
        self._reward_np, self._done_np, self._max_np = b._reward.numpy(), b._done.numpy(), b._max.numpy()
        self._args = (b.boards.data_ptr(), b.aux.data_ptr())
        self._outs = (b._reward.data_ptr(), b._done.data_ptr(), b._max.data_ptr(), b.status.data_ptr())

    def reset(self) -> np.ndarray:
        if self._keep_first:                                         # the constructor's game is the first episode
            self._keep_first = False
            return boards_to_raw(self._board_np)
        self._b.reset()                                              # synchronises
        return boards_to_raw(self._board_np)

    def step(self, action: int):
        action = int(action)
        if not 0 <= action <= 3:
            raise ValueError(f"action {action} outside 0..3")
        b = self._b
        self._keep_first = False
        N.check(b._L.q2048_env_step_ex(
            self._args[0], self._args[1], self._acts.data_ptr() + action, 1, 4, b.seed, b.env_id0,
            b.ctr & 0xFFFFFFFF, b.env_flags, None, self._outs[0], self._outs[1], self._outs[2],
            self._outs[3], _stream(b.device)), "env_step")
        b.ctr += 1
        _sync(b.device)
        return (boards_to_raw(self._board_np), float(self._reward_np[0]), bool(self._done_np[0]),
                1 << int(self._max_np[0]))

    @property
    def score(self) -> int:
        return int(self._b.score.item())
