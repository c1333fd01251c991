"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol the header
declares, host logic (epsilon schedule, sharding, statistics all-reduce over gloo with
world_size 2), loud failure without a device, and no product import of the oracle."""
import ctypes as C
import importlib
import importlib.util
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, REPO


def test_abi_exports_every_declared_symbol(pkg):
    with open(os.path.join(REPO, "include", "q2048.h")) as fh:
        hdr = fh.read()
    declared = set(re.findall(r"\b(q2048_[a-z0-9_]+)\s*\(", hdr))
    assert {"q2048_env_step", "q2048_env_reset", "q2048_q_choose", "q2048_q_update",
            "q2048_fused_rollout", "q2048_q_lookup", "q2048_env_init"} <= declared
    lib = C.CDLL(pkg._native.LIB_PATH)          # loads without a GPU: no compute calls here
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/q2048.h but not exported"
    host = C.CDLL(pkg._native.HOST_LIB_PATH)    # the CPU twin exports the same ABI, symbol for symbol
    for name in sorted(declared):
        assert hasattr(host, name), f"{name} declared in include/q2048.h but not exported by libq2048_host.so"
    assert set(pkg._native._SIGNATURES) == declared
    L = pkg._native.lib()
    assert L.q2048_abi_version() == pkg._native.ABI_VERSION == 7
    assert L.q2048_sizeof_aux() == 16 and L.q2048_sizeof_slot() == 32
    assert L.q2048_strerror(-4).decode().startswith("unsupported")
    # host-side argument validation needs no device
    assert L.q2048_env_init(None, None, 4, 4, 0, 0, None) == -1
    assert L.q2048_env_init(None, None, 4, 6, 0, 0, None) == -4
    assert L.q2048_env_init(None, None, 4, 5, 0, 0, None) == -1            # 5x5 is supported
    assert L.q2048_fused_rollout(None, None, None, 20, 4, 4, 1, 0.5, 0.1, 0.9, 0, 0, 0, 0, None,
                                 None, None, None) == -1
    assert L.q2048_q_choose(16, 99, 16, 4, 4, 0.5, 0, 0, 0, 0, 16, 16, None) == -2  # bad cap_log2
    assert L.q2048_q_choose(16, 20, 16, 4, 4, 1.5, 0, 0, 0, 0, 16, 16, None) == -6  # eps range
    # flag bits outside the ABI are refused by the shipped library (-7), taken by the measurement build
    assert L.q2048_q_choose(16, 20, 16, 4, 4, 0.5, 0, 0, 0, 1 << 9, 16, 16, None) == -7
    assert L.q2048_fused_rollout(16, 16, 16, 20, 4, 4, 1, 0.5, 0.1, 0.9, 0, 0, 0, 1 << 12, None, None, 16,
                                 None) == -7
    assert L.q2048_det_rollout(16, 16, 16, 20, 4, 4, 1, 0.5, 0.1, 0.9, 0, 0, 0, pkg._native.FLAG_NO_LEARN,
                               None, None, 16, 256, 1 << 20, None) == -7
    assert b"flag" in L.q2048_strerror(-7)
    X = pkg._native.load(pkg._native.build_experiments())        # same ABI, same symbols
    for name in sorted(declared):
        assert hasattr(X, name)
    # the deterministic mode's workspace is host arithmetic: pairs twice over, slots, counts, statistics stripes
    small, big = L.q2048_det_workspace_bytes(1, 20), L.q2048_det_workspace_bytes(1 << 20, 32)
    assert 0 < small < 1 << 20 and small % 256 == 0
    assert 36 * (1 << 20) <= big <= 40 * (1 << 20) and big % 256 == 0
    assert L.q2048_det_workspace_bytes(-1, 20) < 0 and L.q2048_det_workspace_bytes(1 << 31, 20) < 0
    # q2048_det_rollout validates before it launches anything (fake, aligned, non-null addresses)
    det = lambda boards, ws, ws_bytes, B=4, n=4, cap=20, eps=0.5: L.q2048_det_rollout(
        boards, 4096, 8192, cap, B, n, 1, eps, 0.1, 0.9, 0, 0, 0, 0, None, None, 12288, ws, ws_bytes, None)
    assert det(None, 16384, 1 << 20) == -1                       # null boards
    assert det(256, None, 1 << 20) == -1                         # null workspace
    assert det(256, 16384 + 8, 1 << 20) == -3                    # workspace not 256-byte aligned
    assert det(256, 16384, 16) == -2                             # workspace too small
    assert det(256, 16384, 1 << 20, n=6) == -4 and det(256, 16384, 1 << 20, cap=3) == -2
    assert det(256, 16384, 1 << 20, eps=1.5) == -6
    assert det(256, 16384, 1 << 20, B=0) == 0                    # nothing to do


def test_bench_traffic_comes_only_from_a_matching_pmc_profile(tmp_path, monkeypatch):
    """roofline.traffic is a committed counter measurement only for the configuration it was taken
    with (the protocol's 64-step launches, the driver's single 20-step launch) AND for the kernel
    sources it was taken on (a later kernel change must not report stale traffic); any
    other run gets null and what the committed passes were taken with."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cfg = {"boards": 1048576, "steps_per_launch": 64, "cap_log2": 32, "board_size": 4, "eps": 0.95, "strict_td": False}
    sha = bench.kernel_sources_sha16()
    assert len(sha) == 16 and sha == bench.kernel_sources_sha16()
    files = []
    for name, S, val, h in (("a.json", 64, 181.0, sha), ("b.json", 20, 194.0, sha), ("c.json", 16, 200.0, "0" * 16)):
        f = tmp_path / name
        f.write_text(json.dumps({"bytes_per_env_step": val, "source": "rocprofv3 --pmc (test)",
                                 "config": dict(cfg, steps_per_launch=S), "kernel_sources_sha16": h}))
        files.append(str(f))
    monkeypatch.setattr(bench, "PMC_TRAFFIC_FILES", files)
    b64, src, other = bench.committed_pmc_traffic(cfg)
    assert b64 == 181.0 and "rocprofv3" in src and other is None
    assert bench.committed_pmc_traffic(dict(cfg, steps_per_launch=20))[0] == 194.0
    none, _, seen = bench.committed_pmc_traffic(dict(cfg, steps_per_launch=16))       # profiled on other sources
    assert none is None and len(seen) == 3 and seen[2]["kernel_sources_sha16"] == "0" * 16
    assert bench.committed_pmc_traffic(dict(cfg, board_size=5))[0] is None
    # whatever is committed under profiles/ says which sources it was taken on
    monkeypatch.undo()
    for path in bench.PMC_TRAFFIC_FILES:
        if os.path.exists(path):
            assert len(json.load(open(path))["kernel_sources_sha16"]) == 16, path


def test_header_structs_match_numpy_layout(pkg):
    assert pkg.AUX_DTYPE.itemsize == 16
    assert [pkg.AUX_DTYPE.fields[k][1] for k in pkg.AUX_DTYPE.names] == [0, 4, 8, 9, 10, 12]


def test_rollout_opts_struct_matches_the_header(pkg, tmp_path):
    """_native.RolloutOpts (ctypes) against q2048_rollout_opts as a C compiler lays it out from
    include/q2048.h: size and every field offset; and the mirror's word count."""
    import ctypes as C
    import subprocess

    N = pkg._native
    fields = [f[0] for f in N.RolloutOpts._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "q2048.h"\nint main(void) {\n'
                   '  printf("%zu %d %d", sizeof(q2048_rollout_opts), Q2048_MIRROR_SEQ, Q2048_MIRROR_WORDS);\n'
                   + "".join(f'  printf(" %zu", offsetof(q2048_rollout_opts, {f}));\n' for f in fields)
                   + "  return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"), "-o", str(exe), str(src)],
                   check=True)
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert got[0] == C.sizeof(N.RolloutOpts) and got[1] == N.MIRROR_SEQ and got[2] == N.MIRROR_WORDS
    assert got[3:] == [getattr(N.RolloutOpts, f).offset for f in fields]
    assert N.RolloutOpts().size == got[0]


def test_product_fails_loudly_without_gpu(pkg):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no HIP device|MI355X"):
        pkg.BatchedGame2048Env(8)                        # the default device is "cuda": no silent move to the CPU twin
    with pytest.raises(RuntimeError, match="no HIP device"):
        pkg.BatchedQLearningAgent(10, device="cuda:0")
    with pytest.raises(RuntimeError):
        pkg.BatchedQLearningAgent(10, device="meta")
    # the product also never reaches for the host library on its own: only lib_for(cpu) loads it
    src = open(os.path.join(REPO, "2048_q-learning_amd", "_native.py")).read()
    assert src.count("load(HOST_LIB_PATH)") == 1 and "def host_lib" in src


def test_product_never_imports_the_oracle():
    pkgdir = os.path.join(REPO, "2048_q-learning_amd")
    for root, _, files in os.walk(pkgdir):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".inc")):
                with open(os.path.join(root, f)) as fh:
                    src = fh.read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
                assert "liboracle" not in src and "q2048_oracle" not in src, f
    for f in ("q2048_host.cpp",):                           # the CPU twin links nothing of the oracle either
        with open(os.path.join(pkgdir, "csrc", f)) as fh:
            src = fh.read()
        assert "orc_" not in src and '#include "q2048_oracle' not in src, f
    for f in ("train.py", "q2048_amd.py"):
        path = os.path.join(REPO, f)
        if os.path.exists(path):
            with open(path) as fh:
                assert not re.search(r"^\s*(from|import)\s+oracle", fh.read(), re.M), f


def test_epsilon_schedule_matches_reference_golden(pkg):
    with open(os.path.join(GOLDEN, "g5_agent.json")) as fh:
        sched = json.load(fh)["epsilon_schedule"]
    for key, want in sched.items():
        E, e0, emin = key.split(",")
        s = pkg.EpsilonSchedule(int(E), float(e0), float(emin))
        got = []
        for ep in range(len(want)):
            s.decay_exploration(ep)
            got.append(s.epsilon)
        assert got == want, key          # bit-identical floats


def test_shard_plan(pkg):
    for total, world in [(8, 8), (10, 3), (1 << 23, 8), (7, 7), (1000, 6)]:
        shards = [pkg.shard_plan(total, world, r, base_env_id=100) for r in range(world)]
        assert sum(s.num_envs for s in shards) == total
        assert shards[0].env_id0 == 100
        for a, b in zip(shards, shards[1:]):
            assert a.env_id0 + a.num_envs == b.env_id0
        assert max(s.num_envs for s in shards) - min(s.num_envs for s in shards) <= 1
    w = pkg.weak_shard(1 << 20, 8, 3)
    assert (w.num_envs, w.env_id0, w.total_envs) == (1 << 20, 3 << 20, 1 << 23)
    with pytest.raises(ValueError):
        pkg.shard_plan(2, 4, 0)
    with pytest.raises(ValueError):
        pkg.shard_plan(8, 2, 2)


def test_board_conversions(pkg):
    rng = np.random.default_rng(0)
    b = rng.integers(0, 12, size=(50, 16)).astype(np.uint8)
    raw = pkg.boards_to_raw(b)
    assert raw.shape == (50, 4, 4) and raw.dtype == np.int64
    assert np.array_equal(pkg.raw_to_boards(raw), b)
    assert pkg.boards_to_raw(np.array([[1, 0, 2] + [0] * 13]))[0, 0].tolist() == [2, 0, 4, 0]
    with pytest.raises(ValueError):
        pkg.raw_to_boards(np.full((4, 4), 3))
    # the scalar twin used by the one-env adapters: arrays and tuples of tuples alike
    env_mod = importlib.import_module("2048_q-learning_amd.env")
    out = np.zeros(16, dtype=np.uint8)
    for k in range(50):
        env_mod.state_to_log2(raw[k], out)
        assert np.array_equal(out, b[k])
        env_mod.state_to_log2(tuple(map(tuple, raw[k].tolist())), out)
        assert np.array_equal(out, b[k])
    with pytest.raises(ValueError):
        env_mod.state_to_log2(np.full((4, 4), 3), out)
    with pytest.raises(ValueError):
        env_mod.state_to_log2(np.zeros((3, 4), dtype=int), out)
    state = tuple(map(tuple, raw[0]))
    assert np.array_equal(pkg.raw_to_boards(np.asarray(state).reshape(1, 4, 4))[0], b[0])


def test_run_summary_has_the_reference_layout(pkg, tmp_path):
    """summary.py: per-episode CSV (Agent/main.py:71-76 schema, numpy rows in the Q-Values
    column as the reference writes them) -> the columns of plots/summary_statistics_cleaned.csv."""
    import csv

    rng = np.random.default_rng(5)
    n = 500
    actions = rng.integers(0, 4, n)
    rewards = rng.normal(-0.3, 0.1, n)
    maxes = 1 << rng.integers(5, 11, n)
    path = tmp_path / "debug_log3.9.csv"
    with open(path, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Episode", "Action", "Q-Values", "Reward", "Total-Reward", "Max Value"])
        for e in range(n):
            w.writerow([e, int(actions[e]), rng.normal(size=4), float(rewards[e]), 12.5, int(maxes[e])])
    row = pkg.summarize_csv(str(path))
    assert pkg.SUMMARY_HEADER == ["Reward_Technique", "Avg_Reward", "Std_Reward", "Max_Value",
                                  "Action_0", "Action_1", "Action_2", "Action_3"]
    assert row[0] == "debug_log3.9" and row[3] == int(maxes.max())
    assert row[1] == pytest.approx(rewards.mean(), rel=1e-12)
    assert row[2] == pytest.approx(rewards.std(ddof=1), rel=1e-12)
    assert row[4:] == np.bincount(actions, minlength=4).tolist() and sum(row[4:]) == n
    out = tmp_path / "summary.csv"
    pkg.write_summary([row, row], str(out))
    lines = out.read_text().strip().splitlines()
    assert lines[0] == ",".join(pkg.SUMMARY_HEADER) and len(lines) == 3
    ref = "/root/reference/QLearningBase/plots/summary_statistics_cleaned.csv"
    if os.path.exists(ref):                       # only where the reference is mounted
        with open(ref) as fh:
            assert fh.readline().strip() == lines[0]
    with pytest.raises(ValueError):
        pkg.summarize_episodes("x", [0, 4], [0.0, 1.0], [2, 4])
    with pytest.raises(ValueError):
        pkg.summarize_episodes("x", [], [], [])
    tool = subprocess.run([sys.executable, os.path.join(REPO, "tools", "summarize_logs.py"), str(path),
                           "-o", str(tmp_path / "s2.csv")], capture_output=True, text=True)
    assert tool.returncode == 0 and tool.stdout.startswith("debug_log3.9,")


_WORKER = r'''
import importlib, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
rank, world = int(sys.argv[2]), int(sys.argv[3])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[4], RANK=str(rank),
                  WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
pkg = importlib.import_module("2048_q-learning_amd")
from oracle import oracle as O                       # tests may use the oracle as the engine
r, lr, w = pkg.dist.init_process_group("gloo")
assert (r, w) == (rank, world)
total, steps, seed = 96, 120, 13
shard = pkg.shard_plan(total, world, rank, base_env_id=500)
envs = O.envs_init(shard.num_envs, 4, seed, shard.env_id0)
agent = O.Agent(100, 4, 0.1, 0.9, 1.0)               # eps = 1: trajectories independent of Q
si, sf = O.rollout(envs, agent, steps, seed, shard.env_id0, 0)
ti, tf = torch.from_numpy(si.copy()), torch.from_numpy(sf.copy())
reducer = pkg.StatsAllReduce(None)                  # the one-collective form bench.py / train.py use
reducer.start(ti, tf)
ti[0] += 1000                                        # after the snapshot
gi, gf = reducer.wait()
ti[0] -= 1000
pkg.allreduce_stats(ti, tf)
slowest = pkg.dist.max_over_ranks(float(rank + 1))
many = pkg.dist.max_over_ranks_many([float(rank), float(10 - rank), 3.5])
pkg.dist.barrier()
np.savez(sys.argv[5] + f".{rank}.npz", boards=envs["board"][:, :16], si=ti.numpy(), sf=tf.numpy(),
         local_si=si, id0=shard.env_id0, slowest=slowest, gi=gi, gf=gf, many=np.array(many))
dist.destroy_process_group()
'''


def test_world_size_2_gloo_sharding_and_stats_allreduce(pkg, O, tmp_path):
    """Two processes, gloo: each rank runs its shard (env ids keyed globally), statistics are
    SUM all-reduced; result == the single-process run over the whole batch."""
    world, port = 2, str(29000 + os.getpid() % 2000)
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    out = str(tmp_path / "out")
    procs = [subprocess.Popen([sys.executable, str(script), REPO, str(r), str(world), port, out],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    for p in procs:
        log, _ = p.communicate(timeout=240)
        assert p.returncode == 0, log.decode()
    parts = [np.load(out + f".{r}.npz") for r in range(world)]
    total, steps, seed = 96, 120, 13
    envs = O.envs_init(total, 4, seed, 500)
    si, sf = O.rollout(envs, O.Agent(100, 4, 0.1, 0.9, 1.0), steps, seed, 500, 0)
    assert np.array_equal(np.concatenate([p["boards"] for p in parts]), envs["board"][:, :16])
    for p in parts:                                   # every rank holds the reduced vector
        for k in (O.ST_STEPS, O.ST_EPISODES, O.ST_VALID, O.ST_SCORE, O.ST_EXPLORE):
            assert p["si"][k] == si[k], k
        assert np.array_equal(p["si"][O.ST_HIST0:], si[O.ST_HIST0:])
        assert np.allclose(p["sf"], sf, rtol=1e-12)
        assert p["slowest"] == 2.0 and p["many"].tolist() == [1.0, 10.0, 3.5]
        # StatsAllReduce: one all-gather, summed on the host in rank order == the two all-reduces
        assert np.array_equal(p["gi"], p["si"]) and np.array_equal(p["gf"], p["sf"])
    assert parts[0]["local_si"][O.ST_STEPS] * 2 == si[O.ST_STEPS]
    assert int(parts[1]["id0"]) == 500 + 48


# ---- rank launcher (bench.py --gpus N without torchrun) -----------------------------------------
_RANK_CHILD = r'''
import json, os, sys, time
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["MASTER_ADDR"] == "127.0.0.1"
mode = sys.argv[1]
if mode == "fail" and rank == world - 1:
    sys.exit(7)
if mode == "fail":
    time.sleep(60)                      # must be terminated by the launcher, not waited for
if mode == "gloo":                      # the ranks really can rendezvous on MASTER_ADDR:PORT
    import torch, torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    total = float(t.item())
    dist.destroy_process_group()
else:
    total = None
print("noise from rank", rank, file=sys.stderr)
print(json.dumps({"rank": rank, "world": world, "port": os.environ["MASTER_PORT"], "sum": total}), flush=True)
'''


def _load_launcher():
    import importlib.util

    spec = importlib.util.spec_from_file_location(
        "q2048_launch_t", os.path.join(REPO, "2048_q-learning_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_rank_launcher_env_relay_and_failure(tmp_path):
    """N children get RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, exactly rank 0's stdout is relayed,
    a failing child makes the parent's result non-zero and the survivors are terminated."""
    import io
    import time

    L = _load_launcher()
    assert "torch" not in L.__dict__                       # importable before anything heavy
    assert not L.inside_a_launch({}) and L.inside_a_launch({"RANK": "0", "WORLD_SIZE": "2"})
    child = tmp_path / "child.py"
    child.write_text(_RANK_CHILD)
    buf = io.StringIO()
    rc = L.launch_ranks([sys.executable, str(child), "ok"], 3, stdout=buf, timeout=60)
    assert rc == 0
    lines = [ln for ln in buf.getvalue().splitlines() if ln.strip()]
    assert len(lines) == 1                                 # ONE line: rank 0's
    rec = json.loads(lines[0])
    assert rec["rank"] == 0 and rec["world"] == 3 and int(rec["port"]) > 0
    buf = io.StringIO()
    rc = L.launch_ranks([sys.executable, str(child), "gloo"], 2, stdout=buf, timeout=120)
    assert rc == 0 and json.loads(buf.getvalue().strip().splitlines()[-1])["sum"] == 3.0   # gloo chats on stdout
    t0 = time.monotonic()
    rc = L.launch_ranks([sys.executable, str(child), "fail"], 2, stdout=io.StringIO(), timeout=50)
    assert rc == 7 and time.monotonic() - t0 < 30          # did not wait for the sleeping rank
    rc = L.launch_ranks([sys.executable, "-c", "import time; time.sleep(30)"], 2,
                        stdout=io.StringIO(), timeout=1.0)
    assert rc == 124


def test_rank_launcher_never_orphans_its_ranks(tmp_path):
    """a launcher that is itself terminated (a test's subprocess timeout, a scheduler) or
    killed outright must not leave ranks behind holding a GPU.  SIGTERM: the handler stops the ranks
    and the launcher exits 128 + 15; SIGKILL: the kernel delivers SIGTERM to the ranks
    (PR_SET_PDEATHSIG)."""
    import signal
    import time

    rank_py = tmp_path / "rank.py"
    rank_py.write_text("import os, sys, time\n"
                       "open(os.path.join(sys.argv[1], 'pid' + os.environ['RANK']), 'w').write(str(os.getpid()))\n"
                       "time.sleep(120)\n")
    parent_py = tmp_path / "parent.py"
    parent_py.write_text(
        "import importlib.util, sys\n"
        f"spec = importlib.util.spec_from_file_location('l', {os.path.join(REPO, '2048_q-learning_amd', 'launch.py')!r})\n"
        "m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)\n"
        f"raise SystemExit(m.launch_ranks([sys.executable, {str(rank_py)!r}, sys.argv[1]], 2, timeout=100))\n")

    def alive(pid):
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            return False
        # a zombie of an already reaped parent is gone for our purposes
        try:
            return open(f"/proc/{pid}/stat").read().split(")")[1].split()[0] != "Z"
        except OSError:
            return False

    for sig, want in ((signal.SIGTERM, 128 + 15), (signal.SIGKILL, -9)):
        d = tmp_path / f"run{int(sig)}"
        d.mkdir()
        p = subprocess.Popen([sys.executable, str(parent_py), str(d)])
        t_end = time.monotonic() + 30
        while time.monotonic() < t_end and not (os.path.exists(d / "pid0") and os.path.exists(d / "pid1")):
            time.sleep(0.05)
        time.sleep(0.2)
        pids = [int(open(d / f"pid{r}").read()) for r in range(2)]
        assert all(alive(x) for x in pids)
        p.send_signal(sig)
        assert p.wait(timeout=20) == want
        t_end = time.monotonic() + 10
        while time.monotonic() < t_end and any(alive(x) for x in pids):
            time.sleep(0.05)
        assert not any(alive(x) for x in pids), f"ranks survived their launcher ({sig})"


def test_bench_self_launches_its_ranks_and_propagates_failure():
    """`python bench.py --gpus 2` with no process group in the environment starts its own ranks;
    on a box without a GPU every rank fails loudly and so does the parent (no hang, no line)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the -m gpu rehearsal")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "4",
                        "--warmup", "1", "--cpu-seconds", "0"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert p.returncode != 0
    assert p.stdout.strip() == ""
    assert "no HIP device" in p.stderr or "MI355X" in p.stderr


def test_statistics_vectors_share_one_buffer(pkg, monkeypatch):
    """agent.new_stats_vectors: the int64 and the float64 vector are the two halves of one allocation,
    which dist._packed recognises (one copy / one clone for both) and separate tensors are not; ranks
    that share a device divide its free memory among them when candidate tables are sized."""
    import importlib

    import torch

    agent_mod = importlib.import_module("2048_q-learning_amd.agent")
    dist_mod = importlib.import_module("2048_q-learning_amd.dist")
    n = pkg._native
    si, sf = agent_mod.new_stats_vectors("cpu")
    assert si.dtype == torch.int64 and sf.dtype == torch.float64
    assert si.numel() == n.NSTAT_I and sf.numel() == n.NSTAT_F
    raw = dist_mod._packed(si, sf)
    assert raw is not None and raw.numel() == 8 * (n.NSTAT_I + n.NSTAT_F)
    si[3] = 7
    sf[1] = 2.5
    assert int(raw[24:32].view(torch.int64)) == 7
    assert float(raw[8 * n.NSTAT_I + 8:8 * n.NSTAT_I + 16].view(torch.float64)) == 2.5
    assert dist_mod._packed(torch.zeros(n.NSTAT_I, dtype=torch.int64), torch.zeros(n.NSTAT_F, dtype=torch.float64)) is None
    assert dist_mod._packed(si.clone(), sf) is None
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    assert agent_mod._ranks_on_this_device() == 4
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    assert agent_mod._ranks_on_this_device() == 1
    monkeypatch.delenv("WORLD_SIZE")
    assert agent_mod._ranks_on_this_device() == 1


def test_stats_allreduce_object_single_process(pkg):
    import torch

    r = pkg.StatsAllReduce(None)
    si = torch.arange(pkg._native.NSTAT_I, dtype=torch.int64)
    sf = torch.ones(pkg._native.NSTAT_F, dtype=torch.float64)
    r.start(si, sf)
    si[0] = 99                                             # the snapshot is what gets reduced
    a, b = r.wait()
    assert a[0] == 0 and a[5] == 5 and b.tolist() == [1.0] * pkg._native.NSTAT_F
    with pytest.raises(RuntimeError):
        r.wait()


def test_rank_pins_itself_to_the_numa_node_of_its_gpu(tmp_path):
    """launch.pin_to_gpu_numa_node: sysfs reads + one sched_setaffinity, in the rank's own start (VERDICT r4 item
    7a).  A fake sysfs tree: two CPU nodes and three GPUs in the KFD topology, GPUs 0 and 1 on NUMA node 0, GPU 2 on
    node 1; *_VISIBLE_DEVICES re-maps the local rank; anything unknown leaves the process alone."""
    spec = importlib.util.spec_from_file_location("q2048_launch_pin", os.path.join(REPO, "2048_q-learning_amd", "launch.py"))
    launch = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(launch)
    mine = sorted(os.sched_getaffinity(0))
    if len(mine) < 2:
        pytest.skip("needs two CPUs")
    half = len(mine) // 2
    node_cpus = {0: mine[:half], 1: mine[half:]}
    sysfs = tmp_path / "sys"
    topo = sysfs / "class" / "kfd" / "kfd" / "topology" / "nodes"
    gpus = [(0x0300, 0), (0x2300, 0), (0xa300, 1)]                     # location_id (bus << 8), NUMA node
    for n in range(2):                                                  # CPU nodes of the topology come first
        (topo / str(n)).mkdir(parents=True)
        (topo / str(n) / "properties").write_text("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    for k, (loc, node) in enumerate(gpus):
        d = topo / str(2 + k)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count 0\nsimd_count 1024\nlocation_id {loc}\ndomain 0\n")
        pci = sysfs / "bus" / "pci" / "devices" / f"0000:{(loc >> 8) & 0xff:02x}:00.0"
        pci.mkdir(parents=True)
        (pci / "numa_node").write_text(f"{node}\n")
    for node, cpus in node_cpus.items():
        d = sysfs / "devices" / "system" / "node" / f"node{node}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(",".join(str(c) for c in cpus) + "\n")
    assert launch.gpu_numa_nodes(str(sysfs)) == [0, 0, 1]
    assert launch._cpulist("0-3,8,10-11") == {0, 1, 2, 3, 8, 10, 11}
    try:
        assert launch.pin_to_gpu_numa_node(2, str(sysfs), environ={}) == (1, node_cpus[1])
        assert sorted(os.sched_getaffinity(0)) == node_cpus[1]
        os.sched_setaffinity(0, mine)
        assert launch.pin_to_gpu_numa_node(0, str(sysfs), environ={"HIP_VISIBLE_DEVICES": "2,0"}) == (1, node_cpus[1])
        os.sched_setaffinity(0, mine)
        assert launch.pin_to_gpu_numa_node(1, str(sysfs), environ={"HIP_VISIBLE_DEVICES": "2,0"}) == (0, node_cpus[0])
        os.sched_setaffinity(0, mine)
        assert launch.pin_to_gpu_numa_node(0, str(sysfs), environ={"ROCR_VISIBLE_DEVICES": "GPU-abc"}) is None
        assert launch.pin_to_gpu_numa_node(7, str(sysfs), environ={}) is None          # no such GPU
        assert launch.pin_to_gpu_numa_node(0, str(tmp_path / "nothing"), environ={}) is None
        assert sorted(os.sched_getaffinity(0)) == mine
    finally:
        os.sched_setaffinity(0, mine)
