#!/usr/bin/env bash
# The default bench command (2^32 slots, 128 GiB) with its table from hipMalloc or mapped from 2 MiB chunks:
# four processes each, alternating, on one box (the fast / slow placement modes show from run to run).
cd "$GRAFT_REPO_ROOT"
BENCH_ARGS=${BENCH_ARGS:---repeats 3}
for r in 1 2 3 4; do
  for pl in plain chunks; do
    timeout -k 10 300 python3 bench.py --cpu-seconds 0 --no-companions $BENCH_ARGS --placement $pl 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$pl', round(d['value']/1e10, 4), round(d['roofline']['frac'], 4), round(d['ms_per_step']*1e3, 2), d['config']['table_placement'])"
  done
done
