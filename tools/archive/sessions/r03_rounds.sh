#!/usr/bin/env bash
# Does the number of wave rounds matter?  6 waves per SIMD x 1024 SIMDs x 64 lanes = 393,216 boards per round.
cd "$GRAFT_REPO_ROOT"
for r in 1 2; do
for B in 786432 1048576 1179648 1572864; do
  timeout -k 10 300 python3 bench.py --cpu-seconds 0 --no-companions --repeats 3 --cap-log2 32 --boards-per-gpu $B 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); B = $B
print(B, round(B / 393216, 2), 'rounds', round(d['value']/1e10, 4), 'e10 steps/s', round(d['ms_per_step']*1e3*1048576/B, 2), 'us per Mi board-steps', round(d['roofline']['avg_launch_ms'], 4))"
done
done
