"""Multi-GPU sharding of the env batch: one process per GPU, no data-path collective.

The reference runs one env in one process (no distributed code).  Envs are independent, so the
batch shards embarrassingly: rank r owns a contiguous range of GLOBAL env ids, and because the
counter RNG is keyed by the global id, every env's trajectory is independent of the world size.
Each GPU keeps its own Q-table replica.  The only collective is the reduction of the episode
statistics vectors (288 bytes, latency-bound: one all-gather, summed on the host) over RCCL
(`nccl` backend on ROCm) or gloo in the CPU tests."""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np
import torch
import torch.distributed as dist

from . import _native as N


@dataclass(frozen=True)
class Shard:
    rank: int
    world_size: int
    env_id0: int     # first global env id of this rank
    num_envs: int    # envs on this rank
    total_envs: int


def shard_plan(total_envs: int, world_size: int, rank: int, base_env_id: int = 0) -> Shard:
    """Contiguous split of [base, base + total) over ranks; the first `total % world` ranks
    take one extra env."""
    if world_size <= 0 or not 0 <= rank < world_size:
        raise ValueError(f"bad rank {rank} / world_size {world_size}")
    if total_envs < world_size:
        raise ValueError("fewer envs than ranks")
    q, r = divmod(total_envs, world_size)
    n = q + (1 if rank < r else 0)
    start = rank * q + min(rank, r)
    return Shard(rank, world_size, base_env_id + start, n, total_envs)


def weak_shard(envs_per_gpu: int, world_size: int, rank: int, base_env_id: int = 0) -> Shard:
    """Weak scaling (BASELINE config 4: 8 x 1,048,576 boards): fixed envs per GPU."""
    return shard_plan(envs_per_gpu * world_size, world_size, rank, base_env_id)


def env_from_torchrun() -> tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1 process = 1 GPU)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend: str | None = None) -> tuple[int, int, int]:
    """Joins the job's process group.  A process that was not started as a rank of a job (no
    WORLD_SIZE in its environment) stays alone; a launched one joins its group even when the job
    has a single rank, so that `torch.distributed.run --nproc-per-node 1` drives exactly the code
    a multi-GPU job runs (RCCL initialisation, the collectives, the side stream) -- the one part
    of the N > 1 path a one-GPU box can exercise.  `nccl` is RCCL on ROCm."""
    rank, local_rank, world = env_from_torchrun()
    launched = "WORLD_SIZE" in os.environ
    if (world > 1 or launched) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # Q2048_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsals on a 1-GPU box)
            backend = os.environ.get("Q2048_DIST_BACKEND") or (
                "nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            dev = torch.device("cuda", local_rank % max(torch.cuda.device_count(), 1))
            torch.cuda.set_device(dev)
            kw["device_id"] = dev
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def allreduce_stats(stats_i: torch.Tensor, stats_f: torch.Tensor, group=None):
    """SUM all-reduce of the statistics vectors in place (every entry is additive: counts,
    sums, histogram bins).  Tensors stay where they are (HBM for nccl, host for gloo)."""
    if stats_i.numel() != N.NSTAT_I or stats_f.numel() != N.NSTAT_F:
        raise ValueError("unexpected statistics vector length")
    if dist.is_available() and dist.is_initialized():        # a 1-rank group still runs the collective
        dist.all_reduce(stats_i, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(stats_f, op=dist.ReduceOp.SUM, group=group)
    return stats_i, stats_f


def _packed(stats_i: torch.Tensor, stats_f: torch.Tensor):
    """One uint8 tensor over both vectors when they are the two halves of one allocation
    (agent.new_stats_vectors), or None."""
    total = (N.NSTAT_I + N.NSTAT_F) * 8
    st = stats_i.untyped_storage()
    if (st.data_ptr() != stats_f.untyped_storage().data_ptr() or st.nbytes() != total
            or stats_i.storage_offset() != 0 or stats_f.storage_offset() != N.NSTAT_I
            or not stats_i.is_contiguous() or not stats_f.is_contiguous()):
        return None
    return torch.empty(0, dtype=torch.uint8, device=stats_i.device).set_(st, 0, (total,))


class StatsAllReduce:
    """The statistics reduction on a stream of its own -- ONE collective: `start`
    snapshots the two vectors as one 288-byte buffer on the caller's stream and hands it to a side
    stream for an all-gather, so the next rollout launch is not ordered behind the collective; `wait`
    returns the SUM over ranks, added up on the host in rank order (int64 and float64 each in its own
    type: exact counts, and a float sum that does not depend on the collective's reduction tree).
    With one process there is no
    collective and nothing to keep off the caller's stream: the buffer goes to pinned host memory by
    one copy on that stream (ordered after the launches before it and before those after it, so it is
    a snapshot all the same).  With CPU tensors (the gloo tests) it is a host all-gather.  One
    reduction in flight at a time."""

    def __init__(self, device=None, group=None):
        self.group = group
        self.device = None if device is None else torch.device(device)
        on_gpu = self.device is not None and self.device.type == "cuda"
        self.side = torch.cuda.Stream(self.device) if on_gpu else None
        self.nbytes = (N.NSTAT_I + N.NSTAT_F) * 8
        self._pending = False
        self._done = None
        self._world = 1
        self._host = None
        self._cpu_parts = None

    def _host_buffer(self, world: int) -> torch.Tensor:
        if self._host is None or self._host.shape[0] < world:
            self._host = torch.zeros((world, self.nbytes), dtype=torch.uint8).pin_memory()
        return self._host

    def _pack(self, stats_i, stats_f) -> torch.Tensor:
        """One uint8 tensor [nbytes]: stats_i | stats_f -- a fresh buffer (the snapshot)."""
        packed = _packed(stats_i, stats_f)
        if packed is not None:
            return packed.clone()                      # both vectors by one copy
        buf = torch.empty(self.nbytes, dtype=torch.uint8, device=stats_i.device)
        buf[:N.NSTAT_I * 8].view(torch.int64).copy_(stats_i)
        buf[N.NSTAT_I * 8:].view(torch.float64).copy_(stats_f)
        return buf

    def start(self, stats_i: torch.Tensor, stats_f: torch.Tensor, snapshot: bool = True) -> None:
        """`snapshot=False`: the caller promises not to touch the vectors (no launch that adds to them) before
        `wait()` returns -- bench.py's timed region ends there anyway -- so the collective reads them in place
        on the caller's stream: no clone, no hand-over to the side stream."""
        if stats_i.numel() != N.NSTAT_I or stats_f.numel() != N.NSTAT_F:
            raise ValueError("unexpected statistics vector length")
        multi = dist.is_available() and dist.is_initialized()    # a 1-rank group still runs the collective
        world = dist.get_world_size(self.group) if multi else 1
        self._world, self._pending = world, True
        if self.side is None:                                    # CPU tensors (gloo tests)
            snap = self._pack(stats_i, stats_f)
            if multi:
                parts = [torch.empty_like(snap) for _ in range(world)]
                dist.all_gather(parts, snap, group=self.group)
            else:
                parts = [snap]
            self._cpu_parts = torch.stack(parts)
            return
        main = torch.cuda.current_stream(self.device)
        host = self._host_buffer(world)
        if not multi:                                  # no collective: copy out on the caller's stream
            packed = _packed(stats_i, stats_f)
            if packed is not None:
                host[0].copy_(packed, non_blocking=True)
            else:
                host[0, :N.NSTAT_I * 8].view(torch.int64).copy_(stats_i, non_blocking=True)
                host[0, N.NSTAT_I * 8:].view(torch.float64).copy_(stats_f, non_blocking=True)
            self._done = torch.cuda.Event()
            self._done.record(main)
            return
        live = None if snapshot else _packed(stats_i, stats_f)
        if live is not None:                           # in place, on the caller's stream
            gathered = torch.empty((world, self.nbytes), dtype=torch.uint8, device=self.device)
            dist.all_gather_into_tensor(gathered.view(-1), live, group=self.group)       # the one collective
            host[:world].copy_(gathered, non_blocking=True)
            self._done = torch.cuda.Event()
            self._done.record(main)
            return
        snap = self._pack(stats_i, stats_f)            # ordered on `main`
        self.side.wait_stream(main)
        with torch.cuda.stream(self.side):
            gathered = torch.empty((world, self.nbytes), dtype=torch.uint8, device=self.device)
            dist.all_gather_into_tensor(gathered.view(-1), snap, group=self.group)   # the one collective
            host[:world].copy_(gathered, non_blocking=True)
            snap.record_stream(self.side)
            gathered.record_stream(self.side)
            self._done = torch.cuda.Event()
            self._done.record(self.side)

    def wait(self):
        """(stats_i, stats_f) of the last `start`, summed over ranks, as host numpy arrays."""
        if not self._pending:
            raise RuntimeError("wait() without start()")
        self._pending = False
        if self._done is not None:
            self._done.synchronize()
            self._done = None
            parts = self._host.numpy()[:self._world]
        else:
            parts, self._cpu_parts = self._cpu_parts.numpy(), None
        ni, nf = N.NSTAT_I * 8, N.NSTAT_F * 8
        si = parts[:, :ni].copy().view(np.int64).sum(axis=0)
        sf = np.zeros(N.NSTAT_F, dtype=np.float64)
        for r in range(parts.shape[0]):                # rank order: one fixed summation order
            sf += parts[r, ni:ni + nf].copy().view(np.float64)
        return si, sf


def max_over_ranks(value: float, device=None, group=None) -> float:
    """MAX all-reduce of one float (bench timing: the slowest rank defines the step time)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def max_over_ranks_many(values, device=None, group=None) -> list:
    """MAX all-reduce of a list of floats in ONE collective (bench.py: every region's wall and kernel
    time, after the last region)."""
    values = [float(v) for v in values]
    if not (dist.is_available() and dist.is_initialized()) or not values:
        return values
    t = torch.tensor(values, dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return [float(x) for x in t.cpu().tolist()]


def barrier(group=None) -> None:
    if dist.is_available() and dist.is_initialized():
        dist.barrier(group=group)


def merge_stats_numpy(parts_i, parts_f):
    """Host-side reference of the all-reduce (tests)."""
    return np.sum(np.stack(parts_i), axis=0), np.sum(np.stack(parts_f), axis=0)
