#!/usr/bin/env bash
# Round 4, session 1: the fused rollout's row cache + statistics mirror.  Parity first, then the
# driver's command A/B (cache on/off, mirror/copy), then launch(S) with and without the cache.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04a; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "fused or four_call or smoke or abi or split or stats_reduction" > $OUT/pytest_subset.log 2>&1; rc=$?
tail -n 15 $OUT/pytest_subset.log
[ $rc -eq 0 ] || exit 1
for rep in 1 2; do
for mode in "" "--no-row-cache" "--stats-by-copy" "--no-row-cache --stats-by-copy"; do
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-companions $mode 2>$OUT/err.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('driver-cmd [$mode]', 'ms_per_step_us', round(d['ms_per_step']*1e3, 2), 'launch_us', round(r['avg_launch_ms']*1e3, 1), 'frac', round(r['frac'], 4), 'regions', d['region_ms'], 'probe', d['config']['table_placement'].get('probe_us'))" | tee -a $OUT/driver_ab.txt || { tail -5 $OUT/err.log; exit 1; }
done
done
for mode in "" "--no-row-cache"; do
for S in 4 8 16 32 64; do
  timeout -k 10 300 python3 bench.py --cpu-seconds 0 --no-companions --repeats 3 --cap-log2 32 --steps 64 --warmup 64 --steps-per-launch $S $mode 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); S = $S; r = d['roofline']
print('[$mode] S', S, 'launch_us', round(r['avg_launch_ms']*1e3, 1), 'per_step_us', round(r['avg_launch_ms']*1e3/S, 2), 'region per step', round(d['ms_per_step']*1e3, 2), 'launches', r['launches'])" | tee -a $OUT/launch_cost.txt
done
done
