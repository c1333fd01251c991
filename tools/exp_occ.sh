#!/usr/bin/env bash
# occupancy variants of the fused kernel (experiment): libs built into gpurun_out/ on the build box
for w in base 4 6 8; do
  if [ "$w" = base ]; then unset Q2048_LIB_PATH; else export Q2048_LIB_PATH=$PWD/tools/variants/libq2048_w$w.so; fi
  echo "== variant $w"
  python - <<'PY'
import gc, importlib, json, os, sys, torch
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("2048_q-learning_amd")
def run(S, eps, bits=0, steps=128):
    B = 1 << 20
    env = pkg.BatchedGame2048Env(B, seed=0, device="cuda:0")
    agent = pkg.BatchedQLearningAgent(1000, learning_rate=0.1, discount_factor=0.99, exploration_rate=eps, capacity_log2=29, device="cuda:0")
    agent.experiment_bits = 0
    def go(n):
        left = n
        while left > 0:
            k = min(S, left); agent.fused_rollout(env, k); left -= k
    go(64); agent.stats(reset=True); agent.experiment_bits = bits; torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); go(steps); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(json.dumps(dict(S=S, eps=eps, bits=bits, us_per_step=round(ms*1e3/steps, 2))), flush=True)
    del env, agent; gc.collect(); torch.cuda.empty_cache()
run(16, 0.95); run(64, 0.95); run(16, 0.01); run(16, 0.95, (4 << 8) | (1 << 12) | (1 << 13))
PY
done
