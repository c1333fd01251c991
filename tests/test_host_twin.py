"""The CPU twin (libq2048_host.so, device "cpu") under the parity tests of the HIP path.

`tests/test_gpu_parity.py` is loaded a second time with its device set to "cpu", and the tests listed below --
the golden fixtures generated from the reference (G2, G4, G5, G8), the exhaustive line table, the oracle
comparisons of the env, the agent, the fused rollout, the deterministic step, the adapters with the reference's
Python types, checkpoints -- run through the SAME Python host code and the SAME C ABI against the host library,
in the CPU suite, with no GPU.  The host library is an explicit device: these tests also check that it is never
what a "cuda" request gets."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

from conftest import REPO

_spec = importlib.util.spec_from_file_location("q2048_parity_on_cpu", os.path.join(REPO, "tests", "test_gpu_parity.py"))
_par = importlib.util.module_from_spec(_spec)
sys.modules["q2048_parity_on_cpu"] = _par
_spec.loader.exec_module(_par)
_par.DEV = "cpu"

ON_CPU = [
    "test_env_init_matches_oracle", "test_golden_g2_moves_on_device", "test_golden_g4_env_step_on_device",
    "test_exhaustive_lines_on_device", "test_stall_sequence_on_device", "test_bad_action_is_rejected_not_masked",
    "test_abi_argument_errors", "test_golden_g8_dqn_env_on_device", "test_env_profiles_fused_and_unfused_match_oracle",
    "test_play_only_never_touches_the_table", "test_env_rollout_matches_oracle", "test_golden_g5_choose_on_device",
    "test_golden_g5_td_on_device", "test_single_env_loop_matches_oracle",
    "test_fused_rollout_matches_oracle_independent_lanes", "test_strict_td_mode_equals_default_on_private_rows",
    "test_fused_equals_unfused_and_split_launches", "test_env_step_to_two_buffers",
    "test_shared_table_pure_exploration_trajectories_exact", "test_sharding_invariance",
    "test_reference_surface_adapters", "test_5x5_env_init_and_rollout_match_oracle",
    "test_5x5_fused_rollout_matches_oracle_independent_lanes", "test_5x5_fused_equals_unfused",
    "test_5x5_shared_table_pure_exploration", "test_table_full_drops_are_counted_not_raised",
    "test_checkpoint_resume_is_bit_exact", "test_export_dict_is_the_reference_table",
    "test_row_tuple_single_env_matches_oracle", "test_episode_log_matches_reference_csv_rows",
    "test_legal_moves_mask", "test_encode_onehot_matches_reference_encoder", "test_tile_overflow_is_reported",
    "test_steps_counter_and_argument_checks", "test_deterministic_mode_batch_edges",
    "test_no_learn_rollout_reads_but_never_writes", "test_rollout_on_a_full_table_stays_bounded",
    "test_closed_key_set_private_rows_match_oracle", "test_closed_key_set_shared_table_drops_match_oracle",
    "test_closed_key_set_deterministic_mode_is_bit_exact", "test_growing_table_freezes_at_its_largest_capacity",
    "test_import_reports_rows_beyond_the_learning_probe_limit", "test_adapters_follow_the_table_policy",
    "test_shared_table_writes_are_legitimate_values", "test_train_save_then_evaluate_scripts",
    "test_line_summaries_of_a_closed_key_set",
]
for _name in ON_CPU:
    globals()[_name.replace("_on_device", "") + "_on_cpu"] = getattr(_par, _name)


def test_the_host_library_is_an_explicit_device_only(pkg):
    N = pkg._native
    assert os.path.samefile(N.HOST_LIB_PATH, os.path.join(os.path.dirname(pkg.__file__), "csrc", "libq2048_host.so"))
    host = N.lib_for(torch.device("cpu"))
    assert host is N.host_lib() and host.q2048_abi_version() == N.ABI_VERSION
    assert host is not N.lib() and N.lib_for(torch.device("cuda")) is N.lib()      # never one for the other
    with pytest.raises(ValueError):
        N.lib_for(torch.device("meta"))
    if not torch.cuda.is_available():                    # a "cuda" request without a GPU raises; it is not served by the host library
        with pytest.raises(RuntimeError):
            pkg.BatchedGame2048Env(4, device="cuda")
    # the device allocator has no host form, and says so
    import ctypes as C
    out = C.c_void_p()
    assert host.q2048_table_alloc(20, 0, C.byref(out)) == -4 and host.q2048_table_reserve(20, 22, 0, C.byref(out)) == -4


def test_host_tables_are_device_tables_byte_for_byte(pkg, O):
    """Same slot layout, same hash, same probe sequence: a table trained by the host library reads back through
    q2048_table_export row for row as the oracle's dict (G6's job: one env, epsilon-greedy, 300 steps), and the
    slot every key sits in is the one the device's probe sequence (home line first, then the next lines) gives."""
    env = pkg.BatchedGame2048Env(1, seed=7, device="cpu")
    agent = pkg.BatchedQLearningAgent(100, learning_rate=0.1, discount_factor=0.99, exploration_rate=0.3,
                                      capacity_log2=12, seed=7, device="cpu")
    agent.fused_rollout(env, 300)
    envs = O.envs_init(1, 4, 7, 0)
    oa = O.Agent(100, 4, 0.1, 0.99, 0.3)
    O.rollout(envs, oa, 300, 7, 0, 0)
    boards, vals = oa.dump()                             # uint8 [R, 16] log2 boards
    keys = (boards.astype(np.uint64) << (4 * np.arange(16, dtype=np.uint64))).sum(axis=1, dtype=np.uint64)
    k, q = agent.export_rows()
    o1, o2 = np.argsort(keys), np.argsort(k)
    assert np.array_equal(keys[o1], k[o2]) and np.allclose(q[o2], vals[o1], rtol=1e-5, atol=1e-6)
    raw = agent.table.numpy().view(np.uint64).reshape(-1, 4)
    occupied = np.nonzero(raw[:, 0])[0]
    assert len(occupied) == len(keys)
    lines = 1 << 10
    for slot in occupied[:64]:                           # every row lies on its key's bucketised sequence
        h = int(pkg_mix64(int(raw[slot, 0])))
        line0, off = (h & 0xFFF) >> 2, h & 3
        pos = (((int(slot) >> 2) - line0) % lines) * 4 + ((int(slot) - off) & 3)
        assert pos < 64, (slot, pos)


def pkg_mix64(x):
    """q2048::mix64 (csrc/q2048_core.hpp), restated for the layout check."""
    m = (1 << 64) - 1
    x = (x * 0x9E3779B97F4A7C15) & m
    x ^= x >> 29
    x = (x * 0xBF58476D1CE4E5B9) & m
    x ^= x >> 32
    return x


def test_train_py_on_the_cpu_device_reproduces_g6(pkg, tmp_path):
    """`train.py --num-envs 1 --device cpu --episodes 30`: the reference's loop body on the drop-in adapters, on
    the CPU twin -- the rows of its log are the reference's rows of G6 (same seed, draws injected into the
    reference when the fixture was made)."""
    import csv
    import subprocess

    g = dict(np.load(os.path.join(REPO, "tests", "golden", "g6_episodes_seed0.npz"), allow_pickle=False))
    assert int(g["env_id0"]) == 0 and int(g["B"]) == 1 and bool(g["decay"])
    E = int(g["E"])
    p = subprocess.run([sys.executable, os.path.join(REPO, "train.py"), "--num-envs", "1", "--device", "cpu",
                        "--episodes", str(E), "--seed", str(int(g["seed"])), "--alpha", repr(float(g["lr"])),
                        "--gamma", repr(float(g["gamma"])), "--epsilon", repr(float(g["eps0"])), "--log", "cpu.csv"],
                       capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr[-2000:]
    rows = list(csv.reader(open(tmp_path / "cpu.csv")))[1:]
    ends = np.flatnonzero(g["dones"])
    assert len(rows) == E == len(ends)
    for e, (row, idx) in enumerate(zip(rows, ends)):     # Episode, Action, Q-Values, Reward, Total-Reward, Max Value
        assert int(row[0]) == e and int(row[1]) == g["actions"][idx] and int(row[5]) == g["maxes"][idx]
        assert np.float32(float(row[3])) == np.float32(g["rewards"][idx])
        assert np.isclose(float(row[4]), g["ep_returns"][e], rtol=1e-5)


def test_train_two_ranks_on_the_cpu_device_equal_one_rank(tmp_path):
    """The N > 1 path end to end without a GPU: `train.py --device cpu --gpus 2` starts its own two ranks (launch.py),
    they join a gloo group, each plays its contiguous shard of the global env ids on its own table replica, and the
    only communication is the statistics reduction.  At epsilon = 1 (actions are draws) the job's rows -- episodes,
    env-steps, mean return, mean score, best tile -- equal those of ONE rank playing all the ids; only the table
    sizes differ (two replicas against one table)."""
    import csv
    import subprocess

    common = ["--device", "cpu", "--epsilon", "1.0", "--max-steps", "64", "--steps-per-launch", "32", "--episodes", "50"]
    run = lambda *a: subprocess.run([sys.executable, os.path.join(REPO, "train.py"), *common, *a],   # noqa: E731
                                    capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    p2 = run("--gpus", "2", "--num-envs", "2048", "--capacity-log2", "20", "--log", "two.csv")
    assert p2.returncode == 0, p2.stderr[-2000:]
    assert (p2.stdout + p2.stderr).count("table check passed") == 2   # both ranks checked their replica (rank 1 reports on stderr)
    p1 = run("--num-envs", "4096", "--capacity-log2", "21", "--log", "one.csv")
    assert p1.returncode == 0, p1.stderr[-2000:]
    two, one = (list(csv.reader(open(tmp_path / f)))[1:] for f in ("two.csv", "one.csv"))
    assert len(two) == len(one) == 2 and int(two[-1][2]) == 4096 * 64
    for a, b in zip(two, one):
        assert a[:4] == b[:4] and a[5:7] == b[5:7]               # epoch, episodes, env-steps, epsilon; score, best tile
        assert abs(float(a[4]) - float(b[4])) <= 1e-3 * abs(float(b[4]))   # mean return: a float sum in another order
        assert int(a[8]) == int(b[8]) == 0                       # no drops


def test_train_resume_of_a_run_with_a_closed_key_set(tmp_path):
    """`train.py --save` / `--resume` ACROSS the freeze (SURVEY 8(f) row 1 x 7.3): a 2^14-slot table that cannot grow
    closes its key set in the first epoch; the run is stopped at epoch 4 and resumed to 8.  In deterministic mode the
    resumed run's report rows and its final table are the uninterrupted run's, bit for bit -- the envs' visit rows
    travel in the checkpoint (agent.state_dict()["visit_rows"], q2048_rowcache_rebind) and the restored table closes
    its key set again before its first launch."""
    import csv
    import subprocess

    common = ["--device", "cpu", "--num-envs", "512", "--steps-per-launch", "32", "--report-every", "1", "--capacity-log2", "14",
              "--deterministic", "--epsilon", "0.9", "--seed", "3", "--episodes", "8"]
    run = lambda *a: subprocess.run([sys.executable, os.path.join(REPO, "train.py"), *common, *a],   # noqa: E731
                                    capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    rows = lambda f: [r[:9] for r in list(csv.reader(open(tmp_path / f)))[1:]]   # noqa: E731  (Steps/s left out)
    pf = run("--save", "full.pt", "--log", "full.csv")
    assert pf.returncode == 0 and "table frozen at step" in pf.stdout, pf.stderr[-2000:]
    p1 = run("--stop-epoch", "4", "--save", "part.pt", "--log", "p1.csv")
    assert p1.returncode == 0, p1.stderr[-2000:]
    p2 = run("--resume", "part.pt", "--save", "resumed.pt", "--log", "p2.csv")
    assert p2.returncode == 0 and "table frozen at step" in p2.stdout and "(frozen)" in p2.stdout, p2.stderr[-2000:]
    full = rows("full.csv")
    assert rows("p1.csv") + rows("p2.csv") == full and int(full[-1][8]) > 0          # ... drops included
    a, b, part = (torch.load(tmp_path / f, map_location="cpu", weights_only=False) for f in ("full.pt", "resumed.pt", "part.pt"))
    assert "visit_rows" in part and len(part["q"]) == len(a["q"]) <= 0.5 * (1 << 14) + 2 * 512 * 32
    oa, ob = np.argsort(a["keys"]), np.argsort(b["keys"])
    assert np.array_equal(a["keys"][oa], b["keys"][ob]) and np.array_equal(a["q"][oa], b["q"][ob])


def test_eight_ranks_on_the_cpu_device_equal_one_rank(tmp_path):
    """BASELINE configs[3]'s partition at its real world size, without GPUs: EIGHT self-launched ranks (launch.py: eight
    fresh children, a gloo group) on the CPU twin, each owning the global env ids [r B, (r + 1) B) and its own table
    replica.
      1. `bench.py --check-shards --device cpu --gpus 8`: the per-chunk hashes of boards + aux over the global ids
         [0, 8 B) equal those of ONE rank playing all 8 B ids -- rank 7's ids start at 7 B, and a trajectory does not
         depend on the sharding.
      2. `train.py --gpus 8 --device cpu` on GROWING replicas: every rank's table grows on its own (each growth
         checked: rows moved == rows created) and passes its end-of-run check, and the job's rows -- episodes,
         env-steps, score, best tile: the 8-way statistics reduction, summed in rank order -- equal the 1-rank run's.
    No scaling curve comes out of this (eight processes share this host's cores): it is the correctness of the
    8-way split, the launcher and the reduction."""
    import csv
    import json
    import subprocess

    env = dict(os.environ, Q2048_HOST_THREADS="1")
    def shards(gpus, per):
        p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--check-shards", "--device", "cpu", "--gpus",
                            str(gpus), "--boards-per-gpu", str(per), "--steps", "24", "--steps-per-launch", "8"],
                           capture_output=True, text=True, timeout=300, cwd=str(tmp_path), env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])

    eight, one = shards(8, 4096), shards(1, 32768)
    assert eight["n_gpus"] == 8 and eight["total_envs"] == one["total_envs"] == 32768
    assert eight["hashes"] == one["hashes"] and len(one["hashes"]) == 8 and len(set(one["hashes"])) == 8
    assert eight["episodes"] == one["episodes"] and eight["env_steps"] == one["env_steps"] == 32768 * 24
    assert eight["status"] == one["status"] == 0

    common = ["--device", "cpu", "--epsilon", "1.0", "--max-steps", "64", "--steps-per-launch", "32", "--episodes", "50",
              "--initial-capacity-log2", "12"]
    run = lambda *a: subprocess.run([sys.executable, os.path.join(REPO, "train.py"), *common, *a],   # noqa: E731
                                    capture_output=True, text=True, timeout=300, cwd=str(tmp_path), env=env)
    p8 = run("--gpus", "8", "--num-envs", "1024", "--log", "eight.csv")
    assert p8.returncode == 0, p8.stderr[-2000:]
    out8 = p8.stdout + p8.stderr
    assert out8.count("table check passed") == 8                 # every rank checked its replica
    assert all(f"[rank {r}] table grew" in out8 for r in range(8))   # ... which grew on its own
    p1 = run("--num-envs", "8192", "--log", "one.csv")
    assert p1.returncode == 0, p1.stderr[-2000:]
    rows8, rows1 = (list(csv.reader(open(tmp_path / f)))[1:] for f in ("eight.csv", "one.csv"))
    assert len(rows8) == len(rows1) == 2 and int(rows8[-1][2]) == 8192 * 64
    for a, b in zip(rows8, rows1):
        assert a[:4] == b[:4] and a[5:7] == b[5:7]               # epoch, episodes, env-steps, epsilon; score, best tile
        assert abs(float(a[4]) - float(b[4])) <= 1e-3 * abs(float(b[4]))   # mean return: a float sum in another order
        assert int(a[8]) == int(b[8]) == 0                       # no drops


@pytest.mark.parametrize("seed", [21, 22])
def test_fuzz_parity_on_the_cpu_device(seed):
    """tests/fuzz_parity.py (the confidence run of the GPU parity check: random geometry, batch, steps, epsilon,
    learning rate, env profile, reset-shaping option, strict TD, random launch splits, the 4-call API mixed with the
    fused rollout in mid-run; private rows, so every env is one reference agent) on the CPU twin: boards and aux
    bit-exact, every Q row within 1e-5 of the oracle's, table size == the oracle's dict sizes."""
    import subprocess

    p = subprocess.run([sys.executable, os.path.join(REPO, "tests", "fuzz_parity.py"), str(seed), "3"],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, FUZZ_DEVICE="cpu"))
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    assert "3 cases passed" in p.stdout


def test_train_on_the_cpu_device_grows_its_table_like_a_fixed_one(tmp_path):
    """`train.py --device cpu` without --capacity-log2: the host table grows (through the ABI alone: export, a larger
    table, import) whenever the load limit is passed, with the growth's own check (rows moved == rows created) and the
    end-of-run check; with one host thread the run is sequential, so the saved table equals the one of the same run
    on a pre-sized table, row for row and bit for bit."""
    import subprocess

    common = ["--device", "cpu", "--num-envs", "1024", "--episodes", "2", "--steps-per-launch", "32", "--seed", "8"]
    run = lambda *a: subprocess.run([sys.executable, os.path.join(REPO, "train.py"), *common, *a], capture_output=True,  # noqa: E731
                                    text=True, timeout=600, cwd=str(tmp_path), env=dict(os.environ, Q2048_HOST_THREADS="1"))
    pg = run("--initial-capacity-log2", "14", "--save", "grown.pt", "--log", "g.csv")
    assert pg.returncode == 0, pg.stderr[-2000:]
    assert pg.stdout.count("table grew") >= 3 and "table check passed" in pg.stdout
    pf = run("--capacity-log2", "21", "--save", "fixed.pt", "--log", "f.csv")
    assert pf.returncode == 0, pf.stderr[-2000:]
    a, b = (torch.load(tmp_path / f, map_location="cpu", weights_only=False) for f in ("grown.pt", "fixed.pt"))
    oa, ob = np.argsort(a["keys"]), np.argsort(b["keys"])
    assert len(a["keys"]) > 100000 and a["capacity_log2"] > 14
    assert np.array_equal(a["keys"][oa], b["keys"][ob]) and np.array_equal(a["q"][oa], b["q"][ob])


def test_c_host_program_drives_the_cpu_twin(pkg, tmp_path):
    """examples/rollout_host_cpu.c: a plain-C host (malloc + q2048_* calls, no HIP, no Python) linked against
    libq2048_host.so reproduces the Python host's run on device "cpu" -- the same ABI, on host memory.  One host
    thread and epsilon = 0.3 on a SHARED table: the run is sequential, so every statistic, the row count and the
    boards are equal exactly; in one call and cut into calls of 16 steps (the same cut on both sides)."""
    import json
    import shutil
    import subprocess

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc is not available")
    exe = str(tmp_path / "rollout_host_cpu")
    libdir = os.path.dirname(pkg._native.HOST_LIB_PATH)
    subprocess.run([gcc, "-std=c11", "-O1", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"),
                    os.path.join(REPO, "examples", "rollout_host_cpu.c"), "-o", exe, "-L", libdir, "-lq2048_host",
                    f"-Wl,-rpath,{libdir}"], check=True)
    B, steps, seed, cap, eps = 3000, 70, 17, 20, 0.3
    keep = os.environ.get("Q2048_HOST_THREADS")
    os.environ["Q2048_HOST_THREADS"] = "1"
    try:
        for per_call in (0, 16):      # (on a shared table the cut is part of the schedule: the same cut on both sides)
            env = pkg.BatchedGame2048Env(B, seed=seed, device="cpu")
            agent = pkg.BatchedQLearningAgent(100, learning_rate=0.1, discount_factor=0.99, exploration_rate=eps,
                                              capacity_log2=cap, seed=seed, device="cpu")
            left = steps
            while left > 0:
                k = min(per_call or steps, left)
                agent.fused_rollout(env, k)
                left -= k
            st = agent.stats()
            out = subprocess.run([exe, str(B), str(steps), str(seed), str(cap), str(eps), str(per_call)], check=True,
                                 capture_output=True, text=True, env=dict(os.environ, Q2048_HOST_THREADS="1")).stdout
            got = json.loads(out.strip().splitlines()[-1])
            for k in ("steps", "episodes", "valid_moves", "score_sum", "inserts", "drops", "explored"):
                assert got[k] == st[k], (per_call, k)
            assert got["rows"] == agent.table_size() and got["status"] == 0
            assert got["board0"] == env.boards[0].tolist()
            assert np.isclose(got["return_sum"], st["return_sum"], rtol=1e-12)
    finally:
        if keep is None:
            os.environ.pop("Q2048_HOST_THREADS", None)
        else:
            os.environ["Q2048_HOST_THREADS"] = keep
