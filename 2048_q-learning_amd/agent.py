"""Tabular Q-learning agent on MI355X: the host-side mirror of QLearningBase/Agent/main.py
(class QLearningAgent, :14-57).

`BatchedQLearningAgent` keeps the reference's constructor arguments and method names
(`choose_action`, `update_q_value`, `decay_exploration`, `q_table[state]`, `epsilon`) over B
states at once, with the Q-table as a device open-addressed hash table; `fused_rollout` is the
throughput entry point (whole loop body of Agent/main.py:91-101 in one launch).
`QLearningAgent` is the one-state adapter with the reference's Python types."""
from __future__ import annotations

import ctypes as C
import os
import warnings
import weakref

import numpy as np
import torch

from . import _native as N
from .env import (BatchedGame2048Env, _host_zeros, _ptr, _require_gpu, _Staging, _stream, raw_to_boards,
                  state_to_log2)


class EpsilonSchedule:
    """QLearningAgent's exploration schedule (Agent/main.py:15-32, 45-57), host scalars.
    Same arithmetic in the same order as the reference, so the values are bit-identical."""

    def __init__(self, total_epochs, exploration_rate=1.0, exploration_min=0.01):
        self.epsilon = exploration_rate                                        # :19
        self.epsilon_min = exploration_min                                     # :20
        self.total_epochs = total_epochs                                       # :22
        self.first_decay_limit = total_epochs * 0.30                           # :25
        self.second_decay_limit = total_epochs * 0.60                          # :26
        self.third_decay_limit = total_epochs * 0.80                           # :27
        self.slow_decay_1 = (exploration_rate - (exploration_min * 1.5)) / self.first_decay_limit
        self.fast_decay = ((exploration_rate - exploration_min) - (exploration_min * 1.5)) / (
            self.second_decay_limit - self.first_decay_limit)                  # :31
        self.slow_decay_2 = (exploration_min * 1.1 - exploration_min) / (
            self.third_decay_limit - self.second_decay_limit)                  # :32

    def decay_exploration(self, current_epoch):
        if current_epoch < self.first_decay_limit:                             # :46
            self.epsilon = max(self.epsilon_min * 1.5, self.epsilon - self.slow_decay_1)
        elif current_epoch < self.second_decay_limit:                          # :49
            self.epsilon = max(self.epsilon_min * 1.1, self.epsilon - self.fast_decay)
        elif current_epoch < self.third_decay_limit:                           # :52
            self.epsilon = max(self.epsilon_min, self.epsilon - self.slow_decay_2)
        else:
            self.epsilon = self.epsilon_min                                    # :57


EPISODE_DTYPE = np.dtype([("env_id", "<u8"), ("episode", "<u4"), ("action", "u1"), ("max_log2", "u1"),
                          ("steps_lo", "<u2"), ("reward", "<f4"), ("total_return", "<f4"),
                          ("score", "<i4"), ("q", "<f4", (4,)), ("reserved", "<u4")])


class EpisodeLog:
    """Device buffer of finished-episode records (q2048_episode): the batched form of the rows
    log_debug_info appends to debug_log.csv (Agent/main.py:59-62)."""

    def __init__(self, capacity: int, device="cuda"):
        self.device = _require_gpu(device)
        self.capacity = int(capacity)
        self.records = torch.zeros((self.capacity, N.SIZEOF_EPISODE), dtype=torch.uint8, device=self.device)
        self.count = torch.zeros(1, dtype=torch.int64, device=self.device)

    def drain(self) -> np.ndarray:
        """Host copy of the records written so far (arrival order), then empties the log.
        `lost` = episodes that finished while the buffer was full."""
        n = int(self.count.item())
        self.lost = max(0, n - self.capacity)
        rec = self.records[: min(n, self.capacity)].cpu().numpy().view(EPISODE_DTYPE).reshape(-1).copy()
        self.count.zero_()
        return rec

    @staticmethod
    def csv_row(rec) -> list:
        """One record in the reference's column order (Agent/main.py:62)."""
        return [int(rec["episode"]), int(rec["action"]), np.asarray(rec["q"], dtype=np.float64),
                float(rec["reward"]), float(rec["total_return"]), 1 << int(rec["max_log2"])]


class _QTableView:
    """`agent.q_table[state]` (Agent/main.py:16,96): state = tuple of 4 tuples of raw tile
    values; returns the 4 Q-values as float64 (zeros when the state was never updated)."""

    def __init__(self, agent: "BatchedQLearningAgent"):
        self._a = weakref.proxy(agent)  # no reference cycle: the table frees with the agent

    def __getitem__(self, state) -> np.ndarray:
        n = self._a.board_size
        b = torch.from_numpy(raw_to_boards(np.asarray(state).reshape(1, n, n))).to(self._a.device)
        return self._a.q_values(b, env_id=self._a.env_id0)[0].double().cpu().numpy()

    def __len__(self) -> int:
        return self._a.table_size()


def _probe_us(table: torch.Tensor, capacity_log2: int, device: torch.device) -> float:
    """us per 2^20 scattered atomics of q2048_table_probe on `table` (contents unchanged)."""
    L, stream, lanes, steps = N.lib(), _stream(device), 1 << 20, 16
    N.check(L.q2048_table_probe(_ptr(table), capacity_log2, lanes, steps, 7, stream), "q2048_table_probe")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(2):
        N.check(L.q2048_table_probe(_ptr(table), capacity_log2, lanes, steps, 1000 + r, stream),
                "q2048_table_probe")
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (2 * steps)


class _ChunkedTable:
    """Owner of one q2048_table_reserve allocation (a table mapped from small physical chunks that may
    grow up to `max_capacity_log2`: `grow`), handed to torch through __cuda_array_interface__: the
    tensor keeps this object alive, and the memory goes back to the device when the last reference is
    dropped."""

    def __init__(self, capacity_log2: int, device: torch.device, chunk_bytes: int = 0,
                 max_capacity_log2: int | None = None, _adopt: int | None = None):
        self.capacity_log2 = int(capacity_log2)
        self.max_capacity_log2 = int(max_capacity_log2 or capacity_log2)
        self.device = device
        if _adopt is None:
            ptr = C.c_void_p()
            with torch.cuda.device(device):
                N.check(N.lib().q2048_table_reserve(self.capacity_log2, self.max_capacity_log2, chunk_bytes,
                                                    C.byref(ptr)), "q2048_table_reserve")
            _adopt = int(ptr.value)
        self.ptr = _adopt
        self.__cuda_array_interface__ = {"shape": (1 << self.capacity_log2, N.SIZEOF_SLOT), "typestr": "|u1",
                                         "data": (self.ptr, False), "version": 2, "strides": None}
        self._finalizer = weakref.finalize(self, _ChunkedTable._free, self.ptr)

    @staticmethod
    def _free(ptr: int) -> None:
        try:
            N.lib().q2048_table_free(ptr)       # synchronises the table's own device, whatever the current one is
        except Exception:                       # interpreter shutdown: the process's memory goes anyway
            pass

    def tensor(self, device: torch.device) -> torch.Tensor:
        t = torch.as_tensor(self, device=device)
        if t.data_ptr() != self.ptr or t.dtype != torch.uint8:
            raise RuntimeError("torch did not adopt the chunked table in place")
        return t

    def grow(self, new_capacity_log2: int, key_words: int, stream) -> "tuple[_ChunkedTable, int]":
        """q2048_table_grow (host-synchronous): the table of the next capacity, every row moved over, this one
        released (its tensor must not be used again).  Returns (owner of the bigger table, rows moved)."""
        ptr, moved = C.c_void_p(), C.c_int64(0)
        N.check(N.lib().q2048_table_grow(self.ptr, self.capacity_log2, int(new_capacity_log2), key_words,
                                         C.byref(ptr), C.byref(moved), stream), "q2048_table_grow")
        self._finalizer.detach()                # the library released this table's chunks itself
        return _ChunkedTable(new_capacity_log2, self.device, max_capacity_log2=self.max_capacity_log2,
                             _adopt=int(ptr.value)), int(moved.value)

    def grow_begin(self, new_capacity_log2: int) -> "_Growth":
        """q2048_table_grow_begin: the library's host thread starts mapping the bigger table; returns at once."""
        return _Growth(self, int(new_capacity_log2))


class _Growth:
    """One q2048_table_grow_begin .. _commit .. _finish (or _abort): a growth off the caller's critical path.
    Keeps the old table's owner alive until the library has released that table."""

    def __init__(self, owner: _ChunkedTable, new_capacity_log2: int):
        self.owner, self.new_capacity_log2 = owner, new_capacity_log2
        self.handle, self.bigger, self.info, self.prepared_ok = C.c_void_p(), None, {}, None
        with torch.cuda.device(owner.device):
            N.check(N.lib().q2048_table_grow_begin(owner.ptr, owner.capacity_log2, new_capacity_log2,
                                                   C.byref(self.handle)), "q2048_table_grow_begin")

    def ready(self) -> bool:
        """True when the next call (commit, or finish after the commit) will not block."""
        return N.lib().q2048_table_grow_poll(self.handle) != N.PENDING   # (a failure is "ready": the next call reports it)

    def wait(self) -> float:
        """q2048_table_grow_wait: blocks until the bigger table is mapped; returns the ms the host thread spent on
        it (a failure is left for `commit` to report)."""
        ms = C.c_double(0.0)
        self.prepared_ok = N.lib().q2048_table_grow_wait(self.handle, C.byref(ms)) == N.OK
        return float(ms.value)

    def commit(self, key_words: int, stream, verify_count: bool = False) -> _ChunkedTable:
        """q2048_table_grow_commit: the move is queued on `stream`; the bigger table is the table from here on."""
        ptr = C.c_void_p()
        with torch.cuda.device(self.owner.device):
            N.check(N.lib().q2048_table_grow_commit(self.handle, key_words, N.GROW_VERIFY_COUNT if verify_count else 0,
                                                    C.byref(ptr), stream), "q2048_table_grow_commit")
        self.bigger = _ChunkedTable(self.new_capacity_log2, self.owner.device,
                                    max_capacity_log2=self.owner.max_capacity_log2, _adopt=int(ptr.value))
        return self.bigger

    def finish(self) -> int:
        """q2048_table_grow_finish: waits for the move, checks it, hands the old table to the library's host
        thread for release.  Returns the rows moved (= the occupied slots of the old table)."""
        moved = C.c_int64(0)
        code = N.lib().q2048_table_grow_finish(self.handle, C.byref(moved))
        if code == N.OK:
            self.owner._finalizer.detach()      # the library releases the old table itself
        N.check(code, "q2048_table_grow_finish")
        return int(moved.value)

    def abort(self) -> None:
        N.check(N.lib().q2048_table_grow_abort(self.handle), "q2048_table_grow_abort")


def _ranks_on_this_device() -> int:
    """How many ranks of this job run on each GPU (1 on a real multi-GPU node)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    return max(1, -(-world // max(torch.cuda.device_count(), 1)))


def place_table(capacity_log2: int, device: torch.device, placement="auto", max_capacity_log2: int | None = None,
                candidates: int | None = None):
    """Allocate the zeroed table where scattered writes run fast.

    How a multi-GiB table's memory was obtained moves the scattered store / atomic rate of the
    rollout by 15-20 % (DESIGN.md 4 "table placement"; loads do not care): a hipMalloc of 8-32 GiB
    comes back in a slow or a fast state depending on where it lands, small ones nearly always slow,
    while the same table mapped from 2 MiB physical chunks (q2048_table_alloc: HIP virtual-memory
    API) measures as fast as one that spans 128 GiB, on every box so far.  `placement`:
      "auto"    tables of 1..64 GiB: "chunks"; smaller ones (cache-resident) and 128 GiB ones (fast as
                they come: 42.4 us per step from hipMalloc, 42.7 from chunks, four processes each on one
                box; a 64 GiB hipMalloc took 44.3 once and 49.4 three times, chunks 43.6-44.3) "plain"
      "chunks"  q2048_table_alloc: 2 MiB physical chunks mapped into one virtual range (the fastest of
                up to four such tables, as many as fit in half of the free memory together)
      "plain"   torch.zeros (caching allocator -> hipMalloc): whatever the device yields
      n (int)   allocate up to n plain candidates at once, time q2048_table_probe on each (~2 ms,
                contents untouched), keep the fastest, release the others
    `max_capacity_log2` ("chunks" only): the table may grow up to that capacity
    (`BatchedQLearningAgent.grow_table`).  `candidates` ("chunks" only): how many tables to map and probe at most
    (default: up to four on a device of one rank, ONE in a multi-rank job -- eight ranks each mapping 4 x 32 GiB at
    start-up are 8 x 1.2 s of driver calls for a <= 7 % effect -- and one for a table that will grow: it is replaced
    at its first growth anyway, and memory released moments before is what makes the next mapping slow).
    Returns (table, report); report["probe_us"] lists the probe's time on every candidate tried."""
    shape = (1 << capacity_log2, N.SIZEOF_SLOT)
    nbytes = N.SIZEOF_SLOT << capacity_log2
    if placement == "auto":
        if (1 << 30) <= nbytes <= (64 << 30):
            try:
                return place_table(capacity_log2, device, "chunks", max_capacity_log2, candidates)
            except (N.NativeError, RuntimeError) as exc:      # no virtual-memory API on this stack, or no room:
                table = torch.zeros(shape, dtype=torch.uint8, device=device)   # the slower kind of table
                return table, {"mode": "plain", "chunks_failed": str(exc)}
        placement = "plain"
    if placement == "plain":
        return torch.zeros(shape, dtype=torch.uint8, device=device), {"mode": "plain"}
    if placement == "chunks":
        # up to four candidates when the device has the room: chunked tables differ less than hipMalloc'd
        # ones, but on some boxes every second one still probes 10-15 % slower than the rest (57-59 against
        # 49-51 us on 2^28 and 2^30 slots); each candidate costs its mapping + zero-fill (~0.3 s for 32 GiB)
        # (ranks that share one device -- rehearsals of a multi-GPU job on a one-GPU box -- share its free memory)
        free, _ = torch.cuda.mem_get_info(device)
        if candidates is None:
            world = int(os.environ.get("WORLD_SIZE", "1"))
            candidates = 1 if (world > 1 or (max_capacity_log2 or capacity_log2) > capacity_log2) else 4
        tries = int(max(1, min(int(candidates), (0.5 * free / _ranks_on_this_device()) // nbytes)))
        tables, times = [], []
        for _ in range(tries):
            try:
                owner = _ChunkedTable(capacity_log2, device, max_capacity_log2=max_capacity_log2)
                t = owner.tensor(device)
                t._q2048_owner = owner                  # how grow_table finds the allocation again
            except N.NativeError:
                if not tables:
                    raise
                break                                   # the memory went elsewhere meanwhile: keep what there is
            tables.append(t)
            times.append(_probe_us(t, capacity_log2, device))
            del t, owner
        chosen = int(np.argmin(times))
        table = tables[chosen]
        del tables
        return table, {"mode": "chunks", "chunk_bytes": 2 << 20, "candidates": tries, "chosen": chosen,
                       "probe_us": [round(x, 2) for x in times]}
    candidates = int(placement)
    if candidates < 1:
        raise ValueError("placement must be 'auto', 'chunks', 'plain' or a candidate count >= 1")
    tables, times = [], []
    for _ in range(candidates):
        try:
            t = torch.zeros(shape, dtype=torch.uint8, device=device)
        except torch.OutOfMemoryError:
            if not tables:
                raise
            break
        tables.append(t)
        times.append(_probe_us(t, capacity_log2, device))
    chosen = int(np.argmin(times))
    table = tables[chosen]
    del tables, t
    torch.cuda.empty_cache()                                         # give the other candidates back
    return table, {"mode": "candidates", "candidates": len(times),
                   "probe_us": [round(x, 2) for x in times], "chosen": chosen}


def auto_capacity_log2(min_rows: int, device, max_log2: int = 33, floor_log2: int = 30,
                       memory_fraction: float = 0.25) -> int:
    """Table size for a run that will create up to `min_rows` rows: load factor <= 0.5, and not
    below 2^`floor_log2` slots (32 GiB) when that fits in `memory_fraction` of the free device
    memory.  Why a floor: a 2^30-slot table mapped from 2 MiB chunks (`place_table`) runs the rollout
    as fast as one that spans 128 GiB (0.331 against 0.327 of the roofline on the driver's bench
    command), a 2^28-slot one 6 % slower even so -- smaller footprints take scattered atomics more
    slowly on this memory system -- and memory is what an MI355X has plenty of.
    Round 2 took half of the device (2^32 slots) for the same speed."""
    need = max(20, int(np.ceil(np.log2(2.0 * max(int(min_rows), 1)))))
    free, _ = torch.cuda.mem_get_info(torch.device(device))
    fit = int(np.floor(np.log2(max(memory_fraction * free / N.SIZEOF_SLOT, 2.0))))
    return min(max(need, min(fit, floor_log2, max_log2)), 40)


class BatchedQLearningAgent:
    """QLearningAgent over a batch.  Constructor arguments as Agent/main.py:15; extra keyword
    arguments size and place the device table.

    capacity_log2   an int: the table has 2**capacity_log2 slots of 32 B, fixed at construction; when an
                    update finds no free slot within the probe limit it is dropped and counted
                    (stats['drops'], status TABLE_FULL) -- never an exception.
                    "auto": a table that GROWS, like the reference's defaultdict (Agent/main.py:16), without
                    stopping the loop.  It starts at 2**initial_capacity_log2 slots ("auto": 2^30 = 32 GiB when
                    that is at most an eighth of the free device memory, else the largest power of two that is,
                    at least 2^20) and, between launches, grows by a factor 2**growth_step_log2 (4) whenever the
                    rows it holds pass `load_limit` of its capacity (0.35: on a pre-filled table the step costs
                    46 us at load 0.1, 57 at 0.36, 69 at 0.51, 87 at 0.61 -- profiles/r05_load_curve_prefilled.jsonl
                    -- and memory is what the device has plenty of): the next table is mapped by the
                    library's host thread while rollouts go on (`prefetch_growth`: started as soon as the current
                    table is in place; else at half the limit), the move of the rows is queued on the stream
                    between two launches, the check (rows moved == rows the kernels created) and the release of
                    the old table happen later, off the critical path (q2048_table_grow_begin / _commit /
                    _finish).  Up to the largest capacity the device has room for next to its predecessor
                    (2^32 slots = 128 GiB on an MI355X); `growths` lists what happened.  No update is dropped in
                    any run that fits the device.  If the device has no virtual-memory API or no room for the
                    first table the agent falls back to ONE fixed plain table (a warning says so).
                    `async_growth=False`: the host-synchronous q2048_table_grow of round 4 instead (same result,
                    bit for bit: tested).
    freeze_load     what a table does when it CANNOT grow any more (a fixed `capacity_log2`, or a growing one at its
                    largest capacity) -- SURVEY 7.3's "stop inserting and count drops", the device counterpart of a
                    defaultdict (Agent/main.py:16) that has no limit: once the table holds `freeze_load` of its
                    capacity in rows (checked between launches with the kernels' own insert counter, so the load ends
                    within one launch of the limit), every launch carries Q2048_FLAG_NO_NEW_ROWS: rows that exist keep
                    learning, a state without a row reads as the zero row the defaultdict would have created, is not
                    created, and its update is dropped and counted (stats['drops']) -- landing in the env's VISIT ROW,
                    which stands in for the missing row while the env stays in that state (so that an invalid move
                    teaches the next greedy choice, as the defaultdict's fresh row would) and ends when it moves on;
                    a warning says so once and `frozen` is True from then on.  Default 0.5: a 1 Mi-board step on a
                    table frozen there costs what the young learning table's costs (4x4: 47.3 us against 45.5; 5x5:
                    61.4 against 61.5), at 0.6 it is
                    65.9 / 85.5 us, at 0.7 101 / 140, at 0.9 823 / 1206 (profiles/r06_load_curve_frozen*.jsonl), and a
                    table driven to load 1.0 takes 12.9 ms (profiles/r05_claim_first_ab.jsonl).  None: never freeze --
                    the table fills up, probes of absent states walk up to 2^10 slots, and updates that find no slot
                    are dropped (status TABLE_FULL).
    independent     every env owns private rows (keys salted with its global id): B independent
                    learners in one table, exactly B reference agents side by side.
    placement       how the table is allocated (`place_table`): "auto", "chunks", "plain" or a count
                    of candidate allocations to probe.  `self.placement` is the report.
    strict_td       update Q[s][a] with a compare-and-swap loop (concurrent updates of one entry
                    serialise) instead of one store (last writer wins).  Same result whenever
                    no two lanes share (s, a); several times slower when many lanes do.
    row_cache       `choose_action` / `update_q_value` hand the row an env read as next_state on to
                    its next call through a device buffer of 32 B per env (q2048_*_cached: what the
                    fused rollout carries in registers), so an update reads one scattered row instead
                    of two.  Used only on a key match, so any calling pattern is correct."""

    def __init__(self, total_epochs, action_space=4, learning_rate=0.1, discount_factor=0.9,
                 exploration_rate=1.0, exploration_min=0.01, capacity_log2: int = 24,
                 device="cuda", seed: int = 0, env_id0: int = 0, independent: bool = False,
                 strict_td: bool = False, board_size: int = 4, placement="auto", row_cache: bool = True,
                 initial_capacity_log2="auto", max_capacity_log2: int | None = None, load_limit: float = 0.35,
                 growth_step_log2: int = 2, prefetch_growth: bool = True, async_growth: bool = True,
                 verify_growth: bool = False, freeze_load: "float | None" = 0.5):
        self.device = _require_gpu(device)
        self._L = N.lib_for(self.device)
        self.on_gpu = self.device.type == "cuda"
        if action_space != 4:
            raise ValueError("the 2048 action space has 4 actions")
        if board_size not in (4, 5):
            raise NotImplementedError("board_size must be 4 (the reference) or 5")
        self.board_size, self.cells = int(board_size), int(board_size) ** 2
        self.growable = capacity_log2 == "auto"
        self.load_limit, self.growths = float(load_limit), []
        self.growth_step_log2, self.prefetch_growth = max(1, int(growth_step_log2)), bool(prefetch_growth)
        self.async_growth, self.verify_growth = bool(async_growth), bool(verify_growth)
        self._growth = self._retiring = None      # a growth being prepared / one committed and not yet finished
        if freeze_load is not None and not 0.05 <= float(freeze_load) <= 0.95:
            raise ValueError("freeze_load must be in [0.05, 0.95], or None (never freeze)")
        self.freeze_load = None if freeze_load is None else float(freeze_load)
        self.frozen, self.frozen_at = False, None  # the key set is closed (Q2048_FLAG_NO_NEW_ROWS on every launch)
        # a 4x4 table with a closed key set carries LINE SUMMARIES (q2048_table_summarise, Q2048_FLAG_LINE_SUMMARY): one
        # request per lookup of an absent state instead of 2.35 -- written by the first launch after the key set closed
        self.line_summaries, self._summarised = True, False
        if self.growable:
            if not 0.05 <= self.load_limit <= 0.9:
                raise ValueError("load_limit must be in [0.05, 0.9]")
            if self.on_gpu:
                free, _ = torch.cuda.mem_get_info(self.device)
                share = free / _ranks_on_this_device()
            else:                                 # the CPU twin: host memory, a quarter of what is available
                share = 0.25 * os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
                self.async_growth = self.prefetch_growth = False    # (grows by export + import: `grow_table`)
                if initial_capacity_log2 == "auto":
                    initial_capacity_log2 = 22
            if initial_capacity_log2 == "auto":   # an eighth of the free memory, at most 2^30 slots (32 GiB)
                initial_capacity_log2 = min(30, max(20, int(np.floor(np.log2(max(share / 8 / N.SIZEOF_SLOT, 2.0))))))
            capacity_log2 = int(initial_capacity_log2)
            # the largest table that fits next to its predecessor while the rows move over -- and next to the retired
            # tables of earlier growths, which are kept until their memory is needed: with fourfold steps everything
            # below the largest table adds up to a third of it, the uneven last step to a little more (1.7 in all)
            fit = int(np.floor(np.log2(max(0.9 * share / (1.7 * N.SIZEOF_SLOT), 16.0))))
            self.max_capacity_log2 = max(capacity_log2, min(int(max_capacity_log2 or 34), fit, 40))
            placement = "chunks" if self.on_gpu else "plain"   # on a GPU growth is a property of the chunk allocator
        else:
            self.max_capacity_log2 = int(capacity_log2)
        if not self.on_gpu:
            placement = "plain"
        if not 4 <= capacity_log2 <= 40:
            raise ValueError("capacity_log2 must be in [4, 40]")
        self.action_space = action_space                                       # :21
        self.lr = learning_rate                                                # :17
        self.gamma = discount_factor                                           # :18
        self.schedule = EpsilonSchedule(total_epochs, exploration_rate, exploration_min)
        self.total_epochs = total_epochs
        self.capacity_log2 = int(capacity_log2)
        self.seed, self.env_id0 = int(seed), int(env_id0)
        self.flags = (N.FLAG_INDEPENDENT if independent else 0) | (N.FLAG_TD_CAS if strict_td else 0)
        self.experiment_bits = 0  # unstable tuning bits OR-ed into fused_rollout's flags
        self.ctr = 0  # choose_action calls so far = counter word of the step draws
        try:
            self.table, self.placement = place_table(self.capacity_log2, self.device, placement,
                                                     self.max_capacity_log2 if self.growable else None)
        except (N.NativeError, RuntimeError) as exc:
            if not self.growable:
                raise
            # no virtual-memory API on this stack, or no room: one fixed table, as large as a quarter of the
            # free memory allows (at most 2^30 slots), from the ordinary allocator -- and say so
            self.growable = False
            self.capacity_log2 = self.max_capacity_log2 = auto_capacity_log2(0, self.device, floor_log2=30)
            warnings.warn(f'capacity_log2="auto": the growing table could not be mapped ({exc}); using a fixed '
                          f"table of 2^{self.capacity_log2} slots instead (updates beyond it are dropped and counted)")
            self.table, self.placement = place_table(self.capacity_log2, self.device, "plain")
            self.placement["growable_failed"] = str(exc)
        # row bookkeeping (growth, `verify_table`): the rows in the table are known EXACTLY at any stream point --
        # the rows counted at the last count / import (`_rows_base`) + the rows the kernels created since
        # (Q2048_ST_INSERTS, cumulative over statistics resets: `_inserts_folded`) -- at the price of one 8-byte
        # synchronising read, which `_room_for` pays only when an upper bound kept on the host (two new rows per
        # env-step launched since the last read) says a threshold may have been passed
        self._rows_base, self._inserts_at_base = 0, 0
        self._inserts_folded, self._inserts_seen, self._steps_unseen = 0, 0, 0
        self._steps_at_read, self._steps_launched, self._row_rate = 0, 0, 1.0
        self._warned_full = False
        self.stats_i, self.stats_f = new_stats_vectors(self.device)
        self.status = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.q_table = _QTableView(self)
        self.row_cache_enabled = bool(row_cache)
        self._row_cache = None
        # statistics mirror of the fused rollout: the launch's last block copies both vectors into pinned
        # host memory the kernels address directly (q2048_rollout_opts.stats_mirror)
        self._mirror = _host_zeros(N.MIRROR_WORDS, torch.int64, self.device)
        self._mirror_np = self._mirror.numpy()
        self._mirror_ticket = torch.zeros(2, dtype=torch.int32, device=self.device)
        self._mirror_launches = 0
        if self.growable and self.async_growth and self.prefetch_growth:
            self._begin_growth()                  # the next table is being mapped while the run sets itself up

    def _cache(self, B: int):
        """The row cache for a batch of B envs (device pointer or None)."""
        if not self.row_cache_enabled:
            return None
        c = self._row_cache
        if c is None or c.shape[0] != B:
            c = self._row_cache = torch.zeros((B, int(self._L.q2048_sizeof_rowcache(self.board_size))),
                                              dtype=torch.uint8, device=self.device)
        return c

    def invalidate_row_cache(self) -> None:
        """After the table was changed by anything but choose_action / update_q_value."""
        if self._row_cache is not None:
            self._row_cache.zero_()

    # -- reference surface ---------------------------------------------------------------
    @property
    def epsilon(self) -> float:
        return self.schedule.epsilon

    @epsilon.setter
    def epsilon(self, v: float):
        self.schedule.epsilon = float(v)

    @property
    def epsilon_min(self) -> float:
        return self.schedule.epsilon_min

    def decay_exploration(self, current_epoch) -> None:
        self.schedule.decay_exploration(current_epoch)                         # :45-57

    def choose_action(self, boards: torch.Tensor) -> torch.Tensor:
        """choose_action (:34-38) for B states -> uint8 actions [B]."""
        boards = self._boards(boards)
        B = boards.shape[0]
        actions = torch.empty(B, dtype=torch.uint8, device=self.device)
        if B == 0:                               # (an empty batch: nothing to choose, and no draw is consumed)
            return actions
        N.check(self._L.q2048_q_choose_cached(
            _ptr(self.table), self.capacity_log2, _ptr(boards), B, self.board_size, float(self.epsilon),
            self.seed, self.env_id0, self.ctr & 0xFFFFFFFF, self._learn_flags(), _ptr(self._cache(B)), _ptr(actions),
            _ptr(self.status), _stream(self.device)), "q_choose")
        self.ctr += 1
        return actions

    def update_q_value(self, boards, actions, reward, next_boards, done) -> None:
        """update_q_value (:40-43) for B transitions."""
        boards, next_boards = self._boards(boards), self._boards(next_boards)
        B = boards.shape[0]
        actions = self._vec(actions, torch.uint8, B, "actions")
        reward = self._vec(reward, torch.float32, B, "reward")
        done = self._vec(done, torch.uint8, B, "done")
        if B == 0:
            return
        self._room_for(B)
        N.check(self._L.q2048_q_update_cached(
            _ptr(self.table), self.capacity_log2, _ptr(boards), _ptr(actions), _ptr(reward),
            _ptr(next_boards), _ptr(done), B, self.board_size, float(self.lr), float(self.gamma), self.env_id0,
            self._learn_flags(), _ptr(self._cache(B)), _ptr(self.stats_i), _ptr(self.status), _stream(self.device)),
            "q_update")

    def q_values(self, boards: torch.Tensor, env_id: int | None = None,
                 return_found: bool = False):
        """q_table[state] for B states -> float32 [B, 4] (zeros where absent).  In independent
        mode board i is looked up in the rows of env `env_id0 + i`, or, when `env_id` is given,
        every board in the rows of that one env."""
        boards = self._boards(boards)
        B = boards.shape[0]
        q = torch.empty((B, 4), dtype=torch.float32, device=self.device)
        found = torch.empty(B, dtype=torch.uint8, device=self.device) if return_found else None
        if B == 0:                               # (an empty batch has no data pointers to hand over)
            return (q, found.bool()) if return_found else q
        flags = self.flags | (N.FLAG_SINGLE_ENV if env_id is not None else 0)
        N.check(self._L.q2048_q_lookup(
            _ptr(self.table), self.capacity_log2, _ptr(boards), B, self.board_size,
            self.env_id0 if env_id is None else int(env_id), flags, _ptr(q), _ptr(found),
            _ptr(self.status), _stream(self.device)), "q_lookup")
        return (q, found.bool()) if return_found else q

    # -- throughput entry point ----------------------------------------------------------
    def fused_rollout(self, env: BatchedGame2048Env, steps: int, episode_log: "EpisodeLog | None" = None,
                      play_only: bool = False, learn: bool = True) -> None:
        """`steps` iterations of choose -> step -> update -> accumulate -> reset-on-done
        (Agent/main.py:91-101, :81) for every env in ONE launch.  Statistics accumulate in
        `stats_i` / `stats_f` on the device (read them with `stats()`); with `episode_log` every
        finished episode also leaves one record (the reference's CSV row, :103-105).  The env's
        profile flags travel with the call.  `play_only`: no learner -- the table is neither read
        nor written (every row reads as zeros); with epsilon = 1 that is uniformly random play.
        `learn=False`: evaluation -- actions come from the stored rows, nothing is created or written."""
        if env.device != self.device:
            raise ValueError("env and agent live on different devices")
        if (env.seed, env.env_id0) != (self.seed, self.env_id0):
            raise ValueError("env and agent must share seed and env_id0 (one draw stream per env)")
        if env.board_size != self.board_size:
            raise ValueError("env and agent have different board sizes")
        if env.ctr != self.ctr:
            raise ValueError(f"env.ctr={env.ctr} and agent.ctr={self.ctr} are out of step")
        log = episode_log
        if learn and not play_only:
            self._room_for(env.num_envs * int(steps))
        # the row every env carries goes from launch to launch (and to choose_action / update_q_value)
        # through the row cache; a learner-less launch touches neither the table nor the cache
        cache = None if play_only else self._cache(env.num_envs)
        opts = N.RolloutOpts(
            log=_ptr(log.records) if log is not None else None, log_capacity=log.capacity if log is not None else 0,
            log_count=_ptr(log.count) if log is not None else None, row_cache=_ptr(cache),
            stats_mirror=self._mirror.data_ptr() if self.stats_i is not None and self.stats_f is not None else None,
            mirror_ticket=_ptr(self._mirror_ticket))
        N.check(self._L.q2048_fused_rollout_opts(
            _ptr(env.boards), _ptr(env.aux), _ptr(self.table), self.capacity_log2, env.num_envs,
            self.board_size, int(steps), float(self.epsilon), float(self.lr), float(self.gamma), self.seed,
            self.env_id0, self.ctr & 0xFFFFFFFF,
            (self._learn_flags() if learn and not play_only else self.flags) | self.experiment_bits | env.env_flags |
            (N.FLAG_PLAY_ONLY if play_only else 0) | (0 if learn else N.FLAG_NO_LEARN),
            _ptr(self.stats_i), _ptr(self.stats_f), _ptr(self.status), C.byref(opts), _stream(self.device)),
            "fused_rollout")
        if env.num_envs > 0 and int(steps) > 0:
            self._mirror_launches += 1
        env.ctr += int(steps)
        self.ctr += int(steps)

    def mirrored_stats(self):
        """(stats_i, stats_f) as the last `fused_rollout` launch left them, read from the host-side
        mirror its last block wrote -- no device call, no copy: valid once the caller has waited for that
        launch (stream / event / device synchronize).  Raises if the mirror is not that launch's."""
        m = self._mirror_np
        if int(m[N.MIRROR_SEQ]) & 0xFFFFFFFF != self._mirror_launches & 0xFFFFFFFF:
            raise RuntimeError(f"statistics mirror holds launch {int(m[N.MIRROR_SEQ])}, expected "
                               f"{self._mirror_launches}: wait for the launch before reading it")
        return m[:N.NSTAT_I].copy(), m[N.NSTAT_I:N.NSTAT_I + N.NSTAT_F].view(np.float64).copy()

    def deterministic_rollout(self, env: BatchedGame2048Env, steps: int) -> None:
        """Reproducible shared-table training (q2048_det_rollout_cached): per step every env acts on the
        table as it is at the start of the step, the updates are sorted on the device by a hash of
        (state, action) and each group is applied update by update in env order.  The result is a
        function of the inputs alone and equals the reference agent (with float32 rows) fed the same
        transitions in env order -- bit for bit, at any B (tested at 1 048 576 boards).  Six launches
        per step, 7.2e9 env-steps/s at 1 Mi boards: `fused_rollout` is the fast path, this one is
        the yard-stick."""
        if env.device != self.device or env.board_size != self.board_size:
            raise ValueError("env and agent do not match")
        if (env.seed, env.env_id0, env.ctr) != (self.seed, self.env_id0, self.ctr):
            raise ValueError("env and agent must share seed, env_id0 and step counter")
        B, L = env.num_envs, self._L
        need = int(L.q2048_det_workspace_bytes(B, self.capacity_log2))
        if need < 0:
            N.check(need, "det_workspace_bytes")
        ws = getattr(self, "_det_ws", None)
        if ws is None or ws.numel() < need + 256:
            ws = self._det_ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
        base = (ws.data_ptr() + 255) & ~255                       # the workspace is 256-byte aligned
        self._room_for(B * int(steps))
        # closed key set: the envs' visit rows live in the row cache from step to step (and come from / go on to
        # fused_rollout and update_q_value in it); otherwise this step neither reads nor keeps the records valid
        cache = self._cache(B) if self.frozen else None
        if cache is None:
            self.invalidate_row_cache()
        N.check(L.q2048_det_rollout_cached(
            _ptr(env.boards), _ptr(env.aux), _ptr(self.table), self.capacity_log2, B, self.board_size,
            int(steps), float(self.epsilon), float(self.lr), float(self.gamma), self.seed, self.env_id0,
            self.ctr & 0xFFFFFFFF, self._learn_flags() | env.env_flags | self.experiment_bits, _ptr(self.stats_i),
            _ptr(self.stats_f), _ptr(self.status), base, need, _ptr(cache), _stream(self.device)), "det_rollout")
        env.ctr += int(steps)
        self.ctr += int(steps)

    # -- a table that grows (capacity_log2="auto") ------------------------------------------------
    def _cumulative_inserts(self) -> int:
        return self._inserts_folded + int(self.stats_i[N.ST_INSERTS].item())

    def _rows_exact(self) -> int:
        """Rows in the table at this point of the stream: one synchronising 8-byte read of the kernels' own
        counter (no scan).  Also refreshes the rows-per-env-step estimate the growth policy plans with."""
        seen = self._cumulative_inserts()
        steps = self._steps_launched - self._steps_at_read
        if steps > 0:
            self._row_rate = min(2.0, max(0.02, 1.25 * (seen - self._inserts_seen) / steps))
        self._inserts_seen, self._steps_at_read, self._steps_unseen = seen, self._steps_launched, 0
        return self._rows_base + seen - self._inserts_at_base

    def _learn_flags(self) -> int:
        """The agent's flags for a call that may create rows: with the key set closed, Q2048_FLAG_NO_NEW_ROWS -- and,
        on a 4x4 table, Q2048_FLAG_LINE_SUMMARY once the summaries of THIS key set are written (one streaming pass,
        queued ahead of the first launch that uses them; a call that may create rows ends their validity)."""
        if not self.frozen:
            self._summarised = False
            return self.flags
        if self.board_size == 4 and self.line_summaries and not self._summarised:
            N.check(self._L.q2048_table_summarise(_ptr(self.table), self.capacity_log2, _stream(self.device)),
                    "table_summarise")
            self._summarised = True
        return self.flags | N.FLAG_NO_NEW_ROWS | (N.FLAG_LINE_SUMMARY if self._summarised and self.line_summaries else 0)

    def _rebase_rows(self, rows: int, table_changed: bool = True) -> None:
        """`rows` rows are in the table now (a count, an import): the new base of the row bookkeeping."""
        if table_changed:
            self.frozen = False                           # (decided again by the next `_room_for`)
            self._summarised = False                      # (line summaries describe a key set that is gone)
        self._rows_base = int(rows)
        self._inserts_at_base = self._inserts_seen = self._cumulative_inserts()
        self._steps_at_read, self._steps_unseen = self._steps_launched, 0

    def _room_for(self, env_steps: int) -> None:
        """Called before anything that may create rows: `env_steps` env-steps are about to be queued, each of
        which creates at most two rows (a state's own and its successor's, Agent/main.py:41-43).  Costs nothing
        while the host-side bound says no threshold can have been passed.  Otherwise the exact row count is read
        (8 bytes, synchronising) and
          - the next table starts being mapped (`prefetch_growth`: already at construction / after each growth;
            else when the rows expected after this call pass half the load limit),
          - the growth is committed -- the move queued between this launch and the previous one -- when the next
            table is ready AND the rows expected after this call (measured rows per env-step x 1.25) pass
            `load_limit` (a quarter of it when the table was prefetched: its memory is committed anyway, and the
            fewer rows there are the cheaper the move); while it is not ready the rollouts go on on the old table
            until the worst case (two rows per env-step) would pass load_limit + 0.25 (at most 0.85), and only
            then wait for it.
        A table that cannot grow (a fixed capacity, or the largest one) closes its key set at `freeze_load`
        (`_freeze_check`); with freeze_load=None it warns once when it passes the load limit."""
        env_steps = int(env_steps)
        self._steps_launched += env_steps
        if self.frozen:
            return
        if self._retiring is not None and self._retiring.ready():
            self._finish_retiring()
        if not self.growable or self.capacity_log2 >= self.max_capacity_log2:
            self._freeze_check(env_steps)
            return
        cap = 1 << self.capacity_log2
        soft, hard = self.load_limit * cap, min(0.85, self.load_limit + 0.25) * cap
        # the load at which this call has something to do: a prefetched table is moved into as soon as it is ready
        # and a quarter of the limit is in use (the fewer rows, the cheaper the move; the memory is committed
        # anyway); without prefetch the mapping begins at half the limit and the move waits for the limit
        commit_at = (0.25 if self.prefetch_growth else 1.0) * soft
        trigger = commit_at if self._growth is not None else min(0.5 * soft, commit_at)
        bound = self._rows_base + self._inserts_seen - self._inserts_at_base + 2 * (self._steps_unseen + env_steps)
        if bound <= trigger or (self._growth is not None and bound <= hard and not self._growth.ready()):
            self._steps_unseen += env_steps              # nothing to decide yet (or nothing to move into yet)
            return
        self._steps_launched -= env_steps                # (the read below must not count steps not yet queued)
        rows = self._rows_exact()
        self._steps_launched += env_steps
        self._steps_unseen = env_steps
        while True:
            cap = 1 << self.capacity_log2
            soft, hard = self.load_limit * cap, min(0.85, self.load_limit + 0.25) * cap
            commit_at = (0.25 if self.prefetch_growth else 1.0) * soft
            expect, worst = rows + self._row_rate * env_steps, rows + 2 * env_steps
            if self.capacity_log2 >= self.max_capacity_log2:   # (the last growth was just committed, or one failed)
                self._freeze_decide(rows)
                return
            if not self.async_growth:
                if expect > soft or worst > hard:
                    self.grow_table(min(self.capacity_log2 + self.growth_step_log2, self.max_capacity_log2), _rows=rows)
                    continue
                return
            if self._growth is None and (self.prefetch_growth or expect > 0.5 * soft):
                self._begin_growth()
            if self._growth is not None and (worst > hard or (expect > commit_at and self._growth.ready())):
                self._commit_growth(rows)
                continue                                  # (a launch larger than the new table's room: again)
            return

    def _freeze_check(self, env_steps: int) -> None:
        """A table that cannot grow, about to take `env_steps` more env-steps.  Free while the host-side bound (two
        new rows per env-step launched since the last read) stays below the threshold; else one synchronising 8-byte
        read of the kernels' insert counter and the decision (`_freeze_decide`)."""
        cap = 1 << self.capacity_log2
        limit = (self.freeze_load if self.freeze_load is not None else self.load_limit) * cap
        bound = self._rows_base + self._inserts_seen - self._inserts_at_base + 2 * (self._steps_unseen + env_steps)
        if bound <= limit or (self.freeze_load is None and self._warned_full):
            self._steps_unseen += env_steps
            return
        self._steps_launched -= env_steps                # (the read must not count steps not yet queued)
        rows = self._rows_exact()
        self._steps_launched += env_steps
        self._steps_unseen = env_steps
        self._freeze_decide(rows)

    def _freeze_decide(self, rows: int) -> None:
        """`rows` rows (exact, at this stream point) in a table that cannot grow: close the key set at freeze_load."""
        cap = 1 << self.capacity_log2
        if self.freeze_load is None:
            if rows > self.load_limit * cap and not self._warned_full:
                self._warned_full = True
                warnings.warn(f"the Q-table is at its largest capacity (2^{self.capacity_log2} slots) and holds "
                              f"{rows} rows (load {rows / cap:.2f} > {self.load_limit}) with freeze_load=None: lookups "
                              "slow down, and updates that find no slot within the probe limit are dropped and counted")
            return
        if rows < self.freeze_load * cap:
            return
        self.frozen = True
        self.frozen_at = {"rows": int(rows), "capacity_log2": self.capacity_log2, "load": rows / cap, "at_step": self.ctr}
        if not self._warned_full:
            self._warned_full = True
            warnings.warn(f"the Q-table cannot grow beyond 2^{self.capacity_log2} slots and holds {rows} rows (load "
                          f"{rows / cap:.2f} >= freeze_load {self.freeze_load}): it takes no new rows from here on -- rows "
                          "that exist keep learning, a state without a row reads as zeros (the defaultdict's fresh row) "
                          "and its updates are dropped and counted (stats()['drops'])")

    def _begin_growth(self) -> None:
        owner = getattr(self.table, "_q2048_owner", None)
        if owner is None or self.capacity_log2 >= self.max_capacity_log2:
            return
        new = min(self.capacity_log2 + self.growth_step_log2, self.max_capacity_log2)
        self._growth = owner.grow_begin(new)
        self._growth.info = {"begun_at_step": self.ctr}

    def _commit_growth(self, rows: int) -> None:
        """Queues the move of the `rows` rows (exact: read at this stream point) into the prepared table."""
        import time
        if self._retiring is not None:                    # the family's counters are free again after its check
            self._finish_retiring()
        g, t0 = self._growth, time.perf_counter()
        ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev[0].record()
        with torch.cuda.device(self.device):              # (raises with the old table intact and still self.table)
            try:
                bigger = g.commit(1 if self.board_size == 4 else 2, _stream(self.device), self.verify_growth)
            except N.NativeError as exc:
                self._growth = None
                if exc.code != N.ERR_ALLOC:
                    raise
                # the device has no room for the next table (other tenants, a smaller card): this capacity is the
                # largest from here on -- the run goes on, and `_room_for` warns when the load passes the limit
                self.max_capacity_log2 = self.capacity_log2
                warnings.warn(f"the Q-table cannot grow beyond 2^{self.capacity_log2} slots: the table of "
                              f"2^{g.new_capacity_log2} slots could not be mapped ({exc})")
                return
        ev[1].record()
        self.table = bigger.tensor(self.device)           # every launch from here on takes the new table
        self.table._q2048_owner = bigger
        g.info.setdefault("prepare_ms", round(g.wait(), 3))  # (ready by now: the commit has returned)
        g.info.update({"from_log2": self.capacity_log2, "to_log2": g.new_capacity_log2, "expected_rows": int(rows),
                       "at_step": self.ctr, "host_ms": round((time.perf_counter() - t0) * 1e3, 3), "events": ev})
        self.capacity_log2 = g.new_capacity_log2
        self.invalidate_row_cache()                       # slots changed
        self._growth, self._retiring = None, g
        if self.prefetch_growth:
            self._begin_growth()

    def _finish_retiring(self) -> None:
        """The committed growth's move is over (or is waited for): rows moved == rows the kernels created, else
        RuntimeError; the old table goes back to the device from the library's host thread."""
        g, self._retiring = self._retiring, None
        moved = g.finish()
        ev = g.info.pop("events")
        ev[1].synchronize()                               # (recorded right behind the library's own event)
        g.info.update({"rows": moved, "ms": round(ev[0].elapsed_time(ev[1]), 3)})
        self.growths.append(g.info)
        if moved != g.info["expected_rows"]:
            raise RuntimeError(f"Q-table self-check failed at the growth 2^{g.info['from_log2']} -> "
                               f"2^{g.info['to_log2']}: {moved} occupied slots moved, {g.info['expected_rows']} rows "
                               "created according to the kernels' counters")

    def wait_for_prefetch(self) -> "float | None":
        """Blocks until the table a prefetched growth is mapping is ready and returns what the library's host thread
        spent on it in ms (None: nothing is being prepared).  For a caller that wants the wait in its set-up rather
        than in its run: hipMemCreate of tens of GiB takes milliseconds on memory that has been free for a while and
        seconds when the driver is still wiping what a process released moments before."""
        if self._growth is None:
            return None
        ms = self._growth.wait()
        self._growth.info["prepare_ms"] = round(ms, 3)
        return ms

    def release_retired(self) -> None:
        """Gives the tables that earlier growths left behind back to the device now (q2048_table_trim): the library
        keeps them until their memory is needed, because freshly released memory makes the next mapping slow."""
        owner = getattr(self.table, "_q2048_owner", None)
        if owner is not None:
            self.finish_growth()
            N.check(self._L.q2048_table_trim(owner.ptr), "q2048_table_trim")

    def finish_growth(self) -> None:
        """Waits for a committed growth's move and runs its check (the end of a run, `verify_table`, tests)."""
        if self._retiring is not None:
            self._finish_retiring()

    def grow_table(self, new_capacity_log2: int | None = None, _rows: int | None = None) -> int:
        """Host-synchronous growth (q2048_table_grow): doubles the table (or takes it to 2**new_capacity_log2
        slots) -- the new capacity mapped onto fresh chunks in an address range of its own, every row moved over
        with one streaming kernel, the new table counted, the smaller one released.  Values are untouched, slots
        (and the table's address) change: the row cache is emptied.  30-180 ms up to 128 GiB on memory that has
        been free for a while.  Only tables made with capacity_log2="auto" can grow.  Returns the
        rows moved.  (`_room_for` uses the asynchronous begin / commit / finish form instead.)"""
        owner = getattr(self.table, "_q2048_owner", None)
        if not self.growable or (owner is None and self.on_gpu):
            raise RuntimeError('only a table made with capacity_log2="auto" can grow')
        new = self.capacity_log2 + 1 if new_capacity_log2 is None else int(new_capacity_log2)
        if not self.capacity_log2 < new <= self.max_capacity_log2:
            raise ValueError(f"cannot grow from 2^{self.capacity_log2} to 2^{new} slots "
                             f"(this table's largest capacity is 2^{self.max_capacity_log2})")
        if not self.on_gpu:
            return self._grow_on_host(new, _rows)
        self.finish_growth()
        if self._growth is not None:                      # a prepared table of another size: give it back
            g, self._growth = self._growth, None
            g.abort()
        expected = self._rows_exact() if _rows is None else int(_rows)
        t0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0[0].record()
        with torch.cuda.device(self.device):     # (raises with the old table intact and still self.table)
            bigger, moved = owner.grow(new, 1 if self.board_size == 4 else 2, _stream(self.device))
        self.table = bigger.tensor(self.device)  # the old tensor's memory is gone: nothing else may hold it
        self.table._q2048_owner = bigger
        t0[1].record()
        t0[1].synchronize()
        self.growths.append({"from_log2": self.capacity_log2, "to_log2": new, "rows": moved, "expected_rows": expected,
                             "ms": round(t0[0].elapsed_time(t0[1]), 3), "at_step": self.ctr})
        self.capacity_log2 = new
        self.invalidate_row_cache()
        if moved != expected:
            raise RuntimeError(f"Q-table self-check failed at the growth to 2^{new}: {moved} occupied slots moved, "
                               f"{expected} rows created according to the kernels' counters")
        return moved

    def _grow_on_host(self, new: int, _rows: int | None) -> int:
        """The CPU twin's growth, through the ABI alone: every row exported, a larger table allocated, every row
        imported (host memory has no chunk allocator to ask; same check: rows moved == rows created)."""
        import time
        expected = self._rows_exact() if _rows is None else int(_rows)
        t0 = time.perf_counter()
        keys, q = self.export_rows()
        self.table = torch.zeros((1 << new, N.SIZEOF_SLOT), dtype=torch.uint8, device=self.device)
        old, self.capacity_log2 = self.capacity_log2, new
        self.invalidate_row_cache()
        if len(q):
            tk = torch.from_numpy(np.ascontiguousarray(keys).view(np.int64).reshape(len(q), -1))
            self.import_rows_device(tk.reshape(-1) if self.board_size == 4 else tk, torch.from_numpy(q))
        if self.check_status() & N.STATUS_TABLE_FULL:
            raise RuntimeError("growth dropped rows (probe limit)")
        moved = self.table_size()
        self.growths.append({"from_log2": old, "to_log2": new, "rows": moved, "expected_rows": expected,
                             "ms": round((time.perf_counter() - t0) * 1e3, 3), "at_step": self.ctr})
        if moved != expected:
            raise RuntimeError(f"Q-table self-check failed at the growth to 2^{new}: {moved} rows moved, {expected} "
                               "rows created according to the counters")
        return moved

    def verify_table(self) -> dict:
        """Run-time check that no row was lost or duplicated: the slots occupied now == the rows counted
        at the last count / import + the rows the kernels say they created since (Q2048_ST_INSERTS).  One
        streaming pass over the table (1.3 ms per 8 GiB), synchronising -- for the end of a run, a checkpoint, a
        test; every growth runs the same check on the table it leaves behind without the extra pass.
        Raises RuntimeError on a mismatch; returns the numbers."""
        self.finish_growth()
        expect = self._rows_exact()
        rows = self.table_size()
        timeouts = N.claim_timeouts(self._L)
        if rows != expect or timeouts:
            raise RuntimeError(f"Q-table self-check failed: {rows} occupied slots, expected {expect} "
                               f"({self._rows_base} counted earlier + {expect - self._rows_base} created since); "
                               f"{timeouts} 5x5 claim time-outs")
        self._rebase_rows(rows, table_changed=False)
        return {"rows": rows, "capacity_log2": self.capacity_log2, "load": rows / float(1 << self.capacity_log2)}

    # -- statistics / table access ---------------------------------------------------------
    def stats(self, reset: bool = False, verify: bool = False) -> dict:
        """Synchronising host copy of the device statistics.  `verify`: also `verify_table()` -- the
        occupied slots must equal the rows counted earlier + the rows created since."""
        si = self.stats_i.cpu().numpy()
        sf = self.stats_f.cpu().numpy()
        if verify:
            self.verify_table()
        if reset:
            self._inserts_folded += int(si[N.ST_INSERTS])
            self.stats_i.zero_()
            self.stats_f.zero_()
        return stats_dict(si, sf)

    def table_size(self) -> int:
        """len(q_table): occupied slots."""
        count = torch.zeros(1, dtype=torch.int64, device=self.device)
        N.check(self._L.q2048_table_count(_ptr(self.table), self.capacity_log2, _ptr(count),
                                          _stream(self.device)), "table_count")
        return int(count.item())

    def export_rows(self):
        """All occupied rows on the host: (keys uint64 [R] for 4x4 / [R, 2] for 5x5,
        q float32 [R, 4])."""
        rows = self.table_size()
        words = 1 if self.board_size == 4 else 2
        keys = torch.empty((max(rows, 1), words), dtype=torch.int64, device=self.device)
        q = torch.empty((max(rows, 1), 4), dtype=torch.float32, device=self.device)
        count = torch.zeros(1, dtype=torch.int64, device=self.device)
        N.check(self._L.q2048_table_export(_ptr(self.table), self.capacity_log2, _ptr(keys),
                                           _ptr(q), rows, words, _ptr(count), _stream(self.device)),
                "table_export")
        got = min(int(count.item()), rows)
        k = keys[:got].cpu().numpy().view(np.uint64)
        return (k[:, 0] if words == 1 else k), q[:got].cpu().numpy()

    def export_dict(self) -> dict:
        """The table in the reference's form: {tuple of 4 tuples of raw tile values ->
        np.float64[4]} (Agent/main.py:16,82).  Shared-table mode only."""
        if self.flags & N.FLAG_INDEPENDENT:
            raise ValueError("salted keys of independent mode do not decode to boards")
        keys, q = self.export_rows()
        out, n = {}, self.board_size
        for k, row in zip(keys.tolist(), q.astype(np.float64)):
            if n == 4:
                cells = [(k >> (4 * c)) & 15 for c in range(16)]
            else:  # 125 bits: key bits 0..62, then reserved bits 0..61
                big = (k[0] & ((1 << 63) - 1)) | ((k[1] & ((1 << 62) - 1)) << 63)
                cells = [(big >> (5 * c)) & 31 for c in range(25)]
            raw = [0 if v == 0 else 1 << v for v in cells]
            out[tuple(tuple(raw[n * r:n * r + n]) for r in range(n))] = row
        return out

    def check_status(self) -> int:
        s = int(self.status.item())
        if s & N.STATUS_BAD_ACTION:
            self.status.zero_()
            raise ValueError("an action outside 0..3 was passed to update_q_value()")
        return s

    # -- checkpoint / resume (the reference never saves its q_table) ------
    def state_dict(self, compact: bool = True) -> dict:
        """Host copy of the learner: the occupied rows (compact) or the raw table, the epsilon
        schedule, counters and statistics.  `load_state_dict` on an agent built with the same
        constructor arguments continues the run; compact rows are re-inserted, so slot positions
        may differ but every lookup returns the same values."""
        sd = {"capacity_log2": self.capacity_log2, "board_size": self.board_size,
              "flags": self.flags, "ctr": self.ctr, "seed": self.seed, "env_id0": self.env_id0,
              "lr": self.lr, "gamma": self.gamma, "schedule": dict(vars(self.schedule)),
              "stats_i": self.stats_i.to("cpu", copy=True), "stats_f": self.stats_f.to("cpu", copy=True)}
        if compact:
            sd["keys"], sd["q"] = self.export_rows()
        else:
            sd["table"] = self.table.to("cpu", copy=True)
        if self.frozen and self._row_cache is not None:
            # closed key set: the envs' visit rows are part of the run (a resumed run equals the uninterrupted one
            # only with them); they live in the row cache, bound to this table's address (q2048_rowcache_rebind)
            sd["visit_rows"] = {"records": self._row_cache.to("cpu", copy=True), "table_address": int(self.table.data_ptr()),
                                "capacity_log2": self.capacity_log2}
        return sd

    def load_state_dict(self, sd: dict) -> None:
        if sd["board_size"] != self.board_size:
            raise ValueError("checkpoint was taken with another board size")
        self.ctr, self.seed, self.env_id0 = int(sd["ctr"]), int(sd["seed"]), int(sd["env_id0"])
        self.lr, self.gamma, self.flags = sd["lr"], sd["gamma"], int(sd["flags"])
        vars(self.schedule).update(sd["schedule"])
        self.stats_i.copy_(sd["stats_i"])
        self.stats_f.copy_(sd["stats_f"])
        self._inserts_folded = 0
        self.finish_growth()
        self.invalidate_row_cache()
        self.table.zero_()
        self._rebase_rows(0)                    # (the restored counters say nothing about this table yet)
        if "table" in sd:
            if sd["capacity_log2"] != self.capacity_log2:
                raise ValueError("raw tables only load into the same capacity")
            self.table.copy_(sd["table"])
        else:
            self.import_rows(sd["keys"], sd["q"])
        self._rebase_rows(self.table_size())
        v = sd.get("visit_rows")
        if v is not None and self.row_cache_enabled:
            # (the key set is the saved one; the launch that follows closes it again -- `_freeze_check` -- before it runs)
            rec = v["records"]
            self._row_cache = rec.to(self.device, copy=True).contiguous()
            N.check(self._L.q2048_rowcache_rebind(_ptr(self._row_cache), rec.shape[0], self.board_size, int(v["table_address"]),
                                                  int(v["capacity_log2"]), _ptr(self.table), self.capacity_log2,
                                                  _stream(self.device)), "rowcache_rebind")

    def import_rows(self, keys: np.ndarray, q: np.ndarray) -> None:
        """Inserts (key, q[4]) rows exported by `export_rows` (any capacity that holds them)."""
        keys = np.ascontiguousarray(keys, dtype=np.uint64).reshape(len(q), -1)
        rows = len(q)
        if rows == 0:
            return
        self.invalidate_row_cache()
        while self.growable and rows * 2 > (1 << self.capacity_log2) and self.capacity_log2 < self.max_capacity_log2:
            self.grow_table(min(self.max_capacity_log2,
                                max(self.capacity_log2 + 1, int(np.ceil(np.log2(2.0 * rows))))))
        # (a table that can grow is taken to load <= 0.5; one that cannot takes what a frozen table of its capacity
        # holds -- freeze_load plus a launch -- and the import's probe limit of 2^14 slots places rows up to load 0.93)
        if rows > 0.9 * (1 << self.capacity_log2):
            raise ValueError("table too small for the checkpoint (load factor would exceed 0.9)")
        tk = torch.from_numpy(keys.view(np.int64)).to(self.device)
        tq = torch.from_numpy(np.ascontiguousarray(q, dtype=np.float32)).to(self.device)
        status = torch.zeros(1, dtype=torch.int32, device=self.device)   # this call's own word: the
        N.check(self._L.q2048_table_import(_ptr(self.table), self.capacity_log2, _ptr(tk), _ptr(tq),  # agent's
                                           rows, keys.shape[1], _ptr(status),                          # is sticky
                                           _stream(self.device)), "table_import")
        code = int(status.item())
        if code & N.STATUS_TABLE_FULL:
            raise RuntimeError("table_import dropped rows (probe limit)")
        if code & N.STATUS_DEEP_ROW:
            warnings.warn("table_import placed rows deeper than the learning paths probe (2^10 slots): q_values finds "
                          "them, choose / update / rollouts read them as absent -- load the checkpoint into a larger table")
        self._rebase_rows(self.table_size())

    def import_rows_device(self, keys: torch.Tensor, q: torch.Tensor) -> None:
        """q2048_table_import of rows that already live on the device: keys int64 [R] (4x4) or [R, 2] (5x5), q
        float32 [R, 4].  No growth, no count: the caller sizes the table and calls `recount_rows()` at the end
        (bulk loads, e.g. bench.py's pre-filled table)."""
        words = 1 if self.board_size == 4 else 2
        rows = int(q.shape[0])
        if keys.dtype != torch.int64 or q.dtype != torch.float32 or keys.numel() != rows * words or q.shape != (rows, 4):
            raise ValueError("keys must be int64 [R] / [R, 2] and q float32 [R, 4]")
        if rows == 0:
            return
        keys, q = keys.to(self.device).contiguous(), q.to(self.device).contiguous()
        self.invalidate_row_cache()
        self._summarised = False                          # (rows arrive: line summaries stop describing the table)
        N.check(self._L.q2048_table_import(_ptr(self.table), self.capacity_log2, _ptr(keys), _ptr(q), rows, words,
                                           _ptr(self.status), _stream(self.device)), "table_import")

    def recount_rows(self) -> int:
        """Counts the occupied slots (one streaming pass, synchronising) and makes that the base of the row
        bookkeeping: after anything but this class's own methods has written rows."""
        self.finish_growth()
        rows = self.table_size()
        self._rebase_rows(rows)
        return rows

    # -- argument plumbing -----------------------------------------------------------------
    def _boards(self, b) -> torch.Tensor:
        if not isinstance(b, torch.Tensor):
            b = torch.as_tensor(np.asarray(b, dtype=np.uint8))
        if b.dtype != torch.uint8:
            raise TypeError("boards must be uint8 log2 tiles")
        b = b.to(self.device)
        if b.dim() != 2 or b.shape[1] != self.cells:
            raise ValueError(f"boards must have shape (B, {self.cells}), got {tuple(b.shape)}")
        return b.contiguous()

    def _vec(self, v, dtype, B, name) -> torch.Tensor:
        if not isinstance(v, torch.Tensor):
            v = torch.as_tensor(v)
        if v.dtype == torch.bool and dtype == torch.uint8:
            v = v.view(torch.uint8)                     # same bytes: no copy, no kernel
        v = v.to(device=self.device, dtype=dtype).contiguous()
        if v.shape != (B,):
            raise ValueError(f"{name} must have shape ({B},), got {tuple(v.shape)}")
        return v


class BatchedRowTupleAgent:
    """BASELINE configs[1]: "flat-array Q over row-tuple features".  The reference's method names
    (choose_action / update_q_value / decay_exploration / epsilon) over a different table: a
    linear Q over the four 16-bit row indices, Q(s,a) = sum_r W[r][idx_r(s)][a], W float32
    [4, 65536, 4] (4 MiB).  It is NOT the reference's whole-board dict learner; it is
    validated against its own restatement in the oracle."""

    def __init__(self, total_epochs, action_space=4, learning_rate=0.1, discount_factor=0.9,
                 exploration_rate=1.0, exploration_min=0.01, device="cuda", seed: int = 0,
                 env_id0: int = 0):
        self.device = _require_gpu(device)
        self._L = N.lib_for(self.device)
        if action_space != 4:
            raise ValueError("the 2048 action space has 4 actions")
        self.lr, self.gamma = learning_rate, discount_factor
        self.schedule = EpsilonSchedule(total_epochs, exploration_rate, exploration_min)
        self.seed, self.env_id0, self.ctr, self.board_size = int(seed), int(env_id0), 0, 4
        self.weights = torch.zeros((4, 65536, 4), dtype=torch.float32, device=self.device)
        self.stats_i, self.stats_f = new_stats_vectors(self.device)
        self.status = torch.zeros(1, dtype=torch.int32, device=self.device)

    epsilon = property(lambda self: self.schedule.epsilon,
                       lambda self, v: setattr(self.schedule, "epsilon", float(v)))

    def decay_exploration(self, current_epoch) -> None:
        self.schedule.decay_exploration(current_epoch)

    def _boards(self, b) -> torch.Tensor:
        if not isinstance(b, torch.Tensor):
            b = torch.as_tensor(np.asarray(b, dtype=np.uint8))
        b = b.to(self.device)
        if b.dtype != torch.uint8 or b.dim() != 2 or b.shape[1] != 16:
            raise ValueError("boards must be uint8 with shape (B, 16)")
        return b.contiguous()

    def choose_action(self, boards) -> torch.Tensor:
        boards = self._boards(boards)
        B = boards.shape[0]
        actions = torch.empty(B, dtype=torch.uint8, device=self.device)
        N.check(self._L.q2048_rt_choose(_ptr(self.weights), _ptr(boards), B, float(self.epsilon),
                                        self.seed, self.env_id0, self.ctr & 0xFFFFFFFF, _ptr(actions),
                                        _stream(self.device)), "rt_choose")
        self.ctr += 1
        return actions

    def update_q_value(self, boards, actions, reward, next_boards, done) -> None:
        boards, next_boards = self._boards(boards), self._boards(next_boards)
        B = boards.shape[0]
        vec = lambda v, dt: torch.as_tensor(v).to(device=self.device, dtype=dt).contiguous()  # noqa: E731
        actions, reward, done = vec(actions, torch.uint8), vec(reward, torch.float32), vec(done, torch.uint8)
        if not (actions.shape == reward.shape == done.shape == (B,)):
            raise ValueError("actions / reward / done must have shape (B,)")
        N.check(self._L.q2048_rt_update(_ptr(self.weights), _ptr(boards), _ptr(actions), _ptr(reward),
                                        _ptr(next_boards), _ptr(done), B, float(self.lr),
                                        float(self.gamma), _ptr(self.status), _stream(self.device)),
                "rt_update")

    def q_values(self, boards) -> torch.Tensor:
        boards = self._boards(boards)
        q = torch.empty((boards.shape[0], 4), dtype=torch.float32, device=self.device)
        N.check(self._L.q2048_rt_lookup(_ptr(self.weights), _ptr(boards), boards.shape[0], _ptr(q),
                                        _stream(self.device)), "rt_lookup")
        return q

    def fused_rollout(self, env: BatchedGame2048Env, steps: int) -> None:
        if env.board_size != 4 or env.device != self.device:
            raise ValueError("the row-tuple learner needs a 4x4 env on the same device")
        if env.env_flags:
            raise ValueError("env profiles are supported by the hash-table agent only")
        if (env.seed, env.env_id0, env.ctr) != (self.seed, self.env_id0, self.ctr):
            raise ValueError("env and agent must share seed, env_id0 and step counter")
        N.check(self._L.q2048_rt_fused_rollout(
            _ptr(env.boards), _ptr(env.aux), _ptr(self.weights), env.num_envs, int(steps),
            float(self.epsilon), float(self.lr), float(self.gamma), self.seed, self.env_id0,
            self.ctr & 0xFFFFFFFF, _ptr(self.stats_i), _ptr(self.stats_f), _ptr(self.status),
            _stream(self.device)), "rt_fused_rollout")
        env.ctr += int(steps)
        self.ctr += int(steps)

    def stats(self, reset: bool = False) -> dict:
        si, sf = self.stats_i.cpu().numpy(), self.stats_f.cpu().numpy()
        if reset:
            self.stats_i.zero_()
            self.stats_f.zero_()
        return stats_dict(si, sf)

    def check_status(self) -> int:
        s = int(self.status.item())
        if s & N.STATUS_BAD_ACTION:
            self.status.zero_()
            raise ValueError("an action outside 0..3 was passed to update_q_value()")
        return s


def new_stats_vectors(device):
    """The two statistics vectors as views of ONE device buffer (int64 part, then float64 part), so
    that a reader (`StatsAllReduce`) can take them with one copy."""
    raw = torch.zeros((N.NSTAT_I + N.NSTAT_F) * 8, dtype=torch.uint8, device=device)
    return raw[:N.NSTAT_I * 8].view(torch.int64), raw[N.NSTAT_I * 8:].view(torch.float64)


def stats_dict(si, sf) -> dict:
    si = np.asarray(si, dtype=np.int64)
    sf = np.asarray(sf, dtype=np.float64)
    ep = int(si[N.ST_EPISODES])
    return {
        "steps": int(si[N.ST_STEPS]), "episodes": ep, "valid_moves": int(si[N.ST_VALID]),
        "score_sum": int(si[N.ST_SCORE]), "inserts": int(si[N.ST_INSERTS]),
        "drops": int(si[N.ST_DROPS]), "explored": int(si[N.ST_EXPLORE]),
        "cas_retries": int(si[N.ST_CAS_RETRY]), "cas_fallbacks": int(si[N.ST_CAS_FALLBACK]),
        "max_tile_hist": {1 << k: int(v) for k, v in enumerate(si[N.ST_HIST0:N.ST_HIST0 + N.ST_HIST_BINS]) if v},
        "return_sum": float(sf[N.SF_RETURN]), "return_sq_sum": float(sf[N.SF_RETURN_SQ]),
        "reward_sum": float(sf[N.SF_REWARD]),
        "mean_return": float(sf[N.SF_RETURN]) / ep if ep else float("nan"),
        "mean_score": float(si[N.ST_SCORE]) / ep if ep else float("nan"),
    }


class _LazyRow:
    """The 4 Q-values of `agent.q_table[state]` (Agent/main.py:96), read back when first used: the
    lookup kernel is queued behind the update it follows and its result is fetched at the next
    synchronisation the loop makes anyway, or on first access.  Behaves like the float64 array the
    reference's dict returns (indexing, len, iteration, numpy conversion, printing)."""

    __slots__ = ("_io", "_off", "_val", "__weakref__")

    def __init__(self, io: _Staging, off: int):
        self._io, self._off, self._val = io, off, None

    def _read(self) -> None:
        if self._val is None:
            self._val = self._io.np[self._off:self._off + 16].view(np.float32).astype(np.float64)

    def _get(self) -> np.ndarray:
        if self._val is None:
            self._io.sync()                      # reads every pending row, this one included
            self._read()
        return self._val

    def __array__(self, dtype=None, copy=None):
        v = self._get()
        return v if dtype is None else v.astype(dtype)

    shape = property(lambda self: (4,))
    dtype = property(lambda self: np.dtype(np.float64))

    def __len__(self):
        return 4

    def __getitem__(self, k):
        return self._get()[k]

    def __iter__(self):
        return iter(self._get())

    def __repr__(self):
        return repr(self._get())

    def __str__(self):
        return str(self._get())

    def __getattr__(self, name):                 # tolist, max, argmax, ... : the array's own
        return getattr(self._get(), name)


class _OneStateTable:
    """`agent.q_table[state]` / `len(agent.q_table)` of the one-state adapter."""

    def __init__(self, owner: "QLearningAgent"):
        self._o = weakref.proxy(owner)

    def __getitem__(self, state) -> _LazyRow:
        o = self._o
        io, b = o._io, o._b
        off = io.take(32)                                        # [0:16] state in, [16:32] row out
        state_to_log2(state, io.np[off:off + 16])
        N.check(b._L.q2048_q_lookup(
            _ptr(b.table), b.capacity_log2, io.ptr + off, 1, 4, b.env_id0, b.flags | N.FLAG_SINGLE_ENV,
            io.ptr + off + 16, None, _ptr(b.status), _stream(b.device)), "q_lookup")
        row = _LazyRow(io, off + 16)
        io.pending.append(weakref.ref(row))
        return row

    def __len__(self) -> int:
        return self._o._b.table_size()


class QLearningAgent:
    """One-state adapter with the reference's exact surface (Agent/main.py:14-57): states are
    tuples of tuples of raw tile values, actions Python ints.  Inputs and outputs travel through a
    ring of pinned host memory that the kernels address directly (`_Staging`): choose_action is
    one launch and one synchronisation, update_q_value and q_table[state] are one launch each."""

    def __init__(self, total_epochs, action_space, learning_rate=0.1, discount_factor=0.9,
                 exploration_rate=1.0, exploration_min=0.01, capacity_log2: int = 22,
                 device="cuda", seed: int = 0, env_id: int = 0):
        self._b = BatchedQLearningAgent(total_epochs, action_space, learning_rate, discount_factor,
                                        exploration_rate, exploration_min, capacity_log2, device,
                                        seed, env_id)
        self._io = _Staging(self._b.device)
        self.q_table = _OneStateTable(self)
        self.action_space = action_space
        self.total_epochs = total_epochs

    lr = property(lambda self: self._b.lr)
    gamma = property(lambda self: self._b.gamma)
    epsilon_min = property(lambda self: self._b.epsilon_min)

    @property
    def epsilon(self) -> float:
        return self._b.epsilon

    @epsilon.setter
    def epsilon(self, v):
        self._b.epsilon = v

    def choose_action(self, state) -> int:
        io, b = self._io, self._b
        off = io.take(32)                                        # [0:16] state in, [16] action out
        state_to_log2(state, io.np[off:off + 16])
        # (the row cache only once the key set is closed: it is what carries the env's visit row from update to choose;
        # before that the plain entry points are one dependent request shorter for a single env)
        N.check(b._L.q2048_q_choose_cached(
            _ptr(b.table), b.capacity_log2, io.ptr + off, 1, 4, float(b.epsilon), b.seed, b.env_id0,
            b.ctr & 0xFFFFFFFF, b._learn_flags(), _ptr(b._cache(1)) if b.frozen else None, io.ptr + off + 16,
            _ptr(b.status), _stream(b.device)), "q_choose")
        b.ctr += 1
        io.sync()
        return int(io.np[off + 16])

    def update_q_value(self, state, action, reward, next_state, done) -> None:
        if not 0 <= int(action) <= 3:
            raise ValueError(f"action {action} outside 0..3")
        io, b = self._io, self._b
        off = io.take(48)             # [0:16] state, [16:32] next state, [32] action, [33] done, [36:40] reward
        buf = io.np
        state_to_log2(state, buf[off:off + 16])
        state_to_log2(next_state, buf[off + 16:off + 32])
        buf[off + 32], buf[off + 33] = int(action), 1 if done else 0
        buf[off + 36:off + 40].view(np.float32)[0] = reward
        b._room_for(1)                # (the table's policy -- growth, or the closed key set of one that cannot grow)
        N.check(b._L.q2048_q_update_cached(
            _ptr(b.table), b.capacity_log2, io.ptr + off, io.ptr + off + 32, io.ptr + off + 36,
            io.ptr + off + 16, io.ptr + off + 33, 1, 4, float(b.lr), float(b.gamma), b.env_id0, b._learn_flags(),
            _ptr(b._cache(1)) if b.frozen else None, _ptr(b.stats_i), _ptr(b.status), _stream(b.device)), "q_update")

    def decay_exploration(self, current_epoch) -> None:
        self._b.decay_exploration(current_epoch)
