#!/usr/bin/env bash
# What one launch of the fused rollout costs on top of its steps: the same 64-step regions cut into
# launches of S steps (HIP-event duration of the launches; 2^32-slot table).
cd "$GRAFT_REPO_ROOT"
for S in 4 8 16 32 64; do
  timeout -k 10 300 python3 bench.py --cpu-seconds 0 --no-companions --repeats 3 --cap-log2 32 --steps 64 --warmup 64 --steps-per-launch $S 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); S = $S; r = d['roofline']
print('S', S, 'launch_us', round(r['avg_launch_ms']*1e3, 1), 'per_step_us', round(r['avg_launch_ms']*1e3/S, 2), 'region per step', round(d['ms_per_step']*1e3, 2), 'launches', r['launches'])"
done
