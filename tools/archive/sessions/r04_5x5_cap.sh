#!/usr/bin/env bash
# 5x5 rollout on the driver's command: 2^30 slots (32 GiB, chunks, best of four candidates) against 2^32 (128 GiB, plain).
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04w; mkdir -p $OUT
for r in 1 2 3; do
  for cap in 30 32; do
    timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-companions --repeats 3 --board-size 5 --cap-log2 $cap 2>>$OUT/err.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('5x5 cap 2^$cap', 'us_per_step', round(d['ms_per_step']*1e3, 2), 'frac', round(r['frac'], 4), d['config']['table_placement'].get('mode'), d['config']['table_placement'].get('probe_us'))" | tee -a $OUT/5x5_cap.txt
  done
done
