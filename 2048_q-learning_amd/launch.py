"""Single-node rank launcher: one fresh process per GPU.

`python bench.py --gpus N` (and train.py) must work when nothing has set up a process group: the
parent then starts N children of the same script, each with the torchrun environment
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), relays rank 0's stdout and returns
the worst exit code.  The reference has no distributed code at all, so there is nothing
it mirrors; the contract is the one `python -m torch.distributed.run --nnodes=1
--nproc-per-node N` gives a script.

Two rules of the GPU pool shape it:
  * a process that has initialised the GPU must never exec another program, so the parent calls
    this BEFORE anything touches HIP (importing torch is fine, `torch.cuda.is_available()` is
    not), and children are brand-new interpreters, never re-executions of the parent;
  * processes are stopped by their exact PID, never by pattern.

This module imports nothing from the package (no torch, nothing that touches HIP): it is safe to
import first.
"""
from __future__ import annotations

import ctypes
import os
import signal
import socket
import subprocess
import sys
import threading
import time

ENV_KEYS = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")


def inside_a_launch(environ=None) -> bool:
    """True when this process already is a rank of a job (torchrun or `launch_ranks`)."""
    environ = os.environ if environ is None else environ
    return "WORLD_SIZE" in environ and "RANK" in environ


def free_port(addr: str = "127.0.0.1") -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind((addr, 0))
        return int(s.getsockname()[1])


def rank_env(rank: int, world_size: int, master_addr: str, master_port: int, base=None) -> dict:
    """The environment of one rank (one node: LOCAL_RANK == RANK)."""
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world_size),
                "LOCAL_WORLD_SIZE": str(world_size), "MASTER_ADDR": master_addr,
                "MASTER_PORT": str(master_port)})
    # dmabuf IPC.  The GPU pool's operating notes: "the host driver only supports dmabuf IPC, and
    # without it RCCL / CUDA-tensor sharing across processes fails with `hipIpcGetMemHandle: invalid
    # argument`"; the pool exports the variable already, so this only matters when a caller hands in
    # an environment built from scratch (`base`).  No N > 1 RCCL run of this repository exists yet
    # (a lease has one GPU), so the setting is carried on the pool's word, not on a measurement.
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _read(path: str):
    try:
        with open(path) as fh:
            return fh.read()
    except OSError:
        return None


def _cpulist(text: str) -> set:
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_nodes(sysfs: str = "/sys") -> list:
    """NUMA node of every GPU, in the order of the KFD topology (= the HIP device order when no
    *_VISIBLE_DEVICES variable re-orders it), read from sysfs alone: no GPU call.  -1 = unknown."""
    root = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
    try:
        ids = sorted(int(d) for d in os.listdir(root) if d.isdigit())
    except OSError:
        return []
    out = []
    for n in ids:
        text = _read(os.path.join(root, str(n), "properties"))
        if text is None:
            continue
        prop = dict(line.split()[:2] for line in text.splitlines() if len(line.split()) >= 2)
        if int(prop.get("simd_count", "0")) == 0:            # a CPU node of the topology
            continue
        loc, dom = int(prop.get("location_id", "0")), int(prop.get("domain", "0"))
        bdf = f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7}"
        node = _read(os.path.join(sysfs, "bus", "pci", "devices", bdf, "numa_node"))
        out.append(int(node) if node is not None and node.strip().lstrip("-").isdigit() else -1)
    return out


def pin_to_gpu_numa_node(local_rank: int, sysfs: str = "/sys", environ=None):
    """Restricts THIS process to the CPUs of the NUMA node its GPU hangs on (`os.sched_setaffinity`): a rank's
    launches, its statistics reads and the library's mapping thread then run next to their device instead of
    across the socket link.  Called by a rank at its own start, before anything touches the GPU -- sysfs reads and
    one affinity call, no re-exec, no child.  A *_VISIBLE_DEVICES list of plain indices is honoured; anything the
    host does not expose (no KFD topology, node -1, an empty intersection with the current affinity) leaves the
    process as it is.  Returns (node, cpus) when it pinned, else None."""
    environ = os.environ if environ is None else environ
    if not hasattr(os, "sched_setaffinity"):
        return None
    nodes = gpu_numa_nodes(sysfs)
    index = int(local_rank)
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        listed = environ.get(var)
        if listed:
            parts = [x.strip() for x in listed.split(",") if x.strip()]
            if not all(x.isdigit() for x in parts) or not parts:
                return None                                   # UUIDs: no mapping without a GPU call
            index = int(parts[index % len(parts)])
            break
    if not nodes or not 0 <= index < len(nodes) or nodes[index] < 0:
        return None
    text = _read(os.path.join(sysfs, "devices", "system", "node", f"node{nodes[index]}", "cpulist"))
    if text is None:
        return None
    cpus = _cpulist(text) & set(os.sched_getaffinity(0))
    if not cpus:
        return None
    os.sched_setaffinity(0, cpus)
    return nodes[index], sorted(cpus)


# libc.prctl, resolved ONCE, here, in the parent: the child must not dlopen between fork and exec (if
# another thread of the launcher held the loader's or malloc's lock at fork time, the child would
# wait for it forever -- the launcher runs under pytest and inside callers that have threads)
try:
    _PRCTL = ctypes.CDLL(None, use_errno=True).prctl
except Exception:                                                       # not Linux: nothing to set
    _PRCTL = None


def _die_with_parent_of(parent_pid: int):
    """preexec_fn of the rank processes (Linux): SIGTERM when the launcher dies, however it dies --
    a rank blocked in a collective would otherwise keep its GPU and its multi-GiB table until the
    c10d timeout.  Between fork and exec it only calls what was resolved beforehand; a launcher that
    died before the prctl took effect is noticed by its changed parent pid."""
    def preexec():
        if _PRCTL is None:
            return
        _PRCTL(1, int(signal.SIGTERM))                                  # PR_SET_PDEATHSIG
        if os.getppid() != parent_pid:                                  # the launcher is gone already
            os.kill(os.getpid(), signal.SIGTERM)
    return preexec


def _stop(procs, grace_s: float = 5.0) -> None:
    """Terminates exactly the PIDs in `procs` that still run: SIGTERM, SIGKILL after `grace_s`."""
    for p in procs:
        if p.poll() is None:
            try:
                p.terminate()
            except OSError:
                pass
    t_end = time.monotonic() + grace_s
    for p in procs:
        try:
            p.wait(max(0.0, t_end - time.monotonic()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()


def launch_ranks(argv, nproc: int, master_addr: str = "127.0.0.1", master_port: int | None = None,
                 timeout: float | None = None, stdout=None, env=None, poll_s: float = 0.05,
                 line_filter=None) -> int:
    """Runs `argv` (a full command line, e.g. [sys.executable, "bench.py", "--gpus", "2"]) as
    `nproc` ranks and waits for them.

    Rank 0's stdout is relayed line by line to `stdout` (default: this process's stdout) -- the
    one JSON line of bench.py; the other ranks' stdout goes to this process's stderr, as does
    every rank's stderr.  `line_filter(line) -> bool` selects which of rank 0's lines are relayed
    (the rest go to stderr: gloo, for one, announces its connections on stdout).  When a rank exits
    non-zero, `timeout` seconds pass, or the launcher itself receives SIGTERM / SIGINT, the remaining
    ranks are terminated (SIGTERM to their PIDs, SIGKILL after 5 s) and the result is non-zero; the
    ranks also get SIGTERM from the kernel if the launcher dies without running its handlers
    (PR_SET_PDEATHSIG).  Returns 0 iff every rank exited 0."""
    if nproc < 1:
        raise ValueError("nproc must be >= 1")
    out = sys.stdout if stdout is None else stdout
    port = free_port(master_addr) if master_port is None else int(master_port)
    procs = []
    worst, failed = 0, False

    class _Signalled(Exception):
        pass

    def on_signal(signum, _frame):
        raise _Signalled(signum)

    # SIGTERM / SIGINT to the launcher (a test's subprocess timeout, a scheduler, ^C) must not orphan
    # the ranks: the handler unwinds into the finally below, which stops them
    old_handlers = {}
    if threading.current_thread() is threading.main_thread():
        for sig in (signal.SIGTERM, signal.SIGINT):
            old_handlers[sig] = signal.signal(sig, on_signal)
    pump = None
    try:
        for r in range(nproc):
            procs.append(subprocess.Popen(
                list(argv), env=rank_env(r, nproc, master_addr, port, env),
                stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr, text=True,
                preexec_fn=_die_with_parent_of(os.getpid())))

        def relay():
            for line in procs[0].stdout:
                dst = out if line_filter is None or line_filter(line) else sys.stderr
                dst.write(line)
                dst.flush()

        pump = threading.Thread(target=relay, daemon=True)
        pump.start()
        deadline = None if timeout is None else time.monotonic() + timeout
        while True:
            codes = [p.poll() for p in procs]
            if any(c not in (None, 0) for c in codes):
                failed = True
                worst = next(c for c in codes if c not in (None, 0))
                break
            if all(c == 0 for c in codes):
                break
            if deadline is not None and time.monotonic() > deadline:
                failed, worst = True, 124
                break
            time.sleep(poll_s)
    except _Signalled as sig:
        failed, worst = True, 128 + int(sig.args[0])
    finally:
        # a second SIGTERM / SIGINT while the ranks are being stopped must not abort the stopping: the
        # signals are ignored for its duration, then the caller's handlers come back
        for sig in old_handlers:
            signal.signal(sig, signal.SIG_IGN)
        try:
            _stop(procs)                      # exactly the PIDs started above; a no-op when all have exited
        finally:
            for sig, h in old_handlers.items():
                signal.signal(sig, h)
    if pump is not None:
        pump.join(timeout=5.0)
    if failed:
        return worst if worst > 0 else 1      # a signal's negative code is still a failure
    return 0
