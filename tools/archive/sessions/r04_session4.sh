#!/usr/bin/env bash
# Round 4, session 4: is the per-block statistics flush part of a launch's fixed cost?  (intercept with the
# statistics pointers NULL); what the host spends before the launch is submitted.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04d; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 500 python3 tools/archive/exp_intercept.py 2>$OUT/err.log | tee $OUT/intercept.jsonl || { tail -5 $OUT/err.log; exit 1; }
timeout -k 10 300 python3 - <<'PY' 2>>$OUT/err.log | tee $OUT/host_cost.txt
import importlib, time, torch, statistics
pkg = importlib.import_module("2048_q-learning_amd")
dev = torch.device("cuda:0")
env = pkg.BatchedGame2048Env(1 << 20, seed=0, device=dev)
agent = pkg.BatchedQLearningAgent(1000, exploration_rate=0.95, capacity_log2=30, seed=0, device=dev)
agent.fused_rollout(env, 64, play_only=True); agent.ctr = env.ctr
agent.fused_rollout(env, 5); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
rec0, call, rec1, sync, read = [], [], [], [], []
for _ in range(9):
    torch.cuda.synchronize()
    t = time.perf_counter(); e0.record(); t1 = time.perf_counter()
    agent.fused_rollout(env, 20); t2 = time.perf_counter()
    e1.record(); t3 = time.perf_counter()
    torch.cuda.synchronize(dev); t4 = time.perf_counter()
    agent.mirrored_stats(); t5 = time.perf_counter()
    rec0.append(t1 - t); call.append(t2 - t1); rec1.append(t3 - t2); sync.append(t4 - t3); read.append(t5 - t4)
    kern = e0.elapsed_time(e1) * 1e3
    print(f"wall {1e6*(t5-t):.1f} us, events {kern:.1f} us, outside {1e6*(t5-t)-kern:.1f}")
med = lambda v: round(statistics.median(v) * 1e6, 1)
print("host us (median): event record", med(rec0), "| fused_rollout call", med(call), "| event record", med(rec1), "| synchronize (incl. the kernel)", med(sync), "| mirror read", med(read))
PY
