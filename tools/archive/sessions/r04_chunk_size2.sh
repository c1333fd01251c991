#!/usr/bin/env bash
# Chunk size against the placement lottery: six fresh allocations (processes) per chunk size, 16 and 32 GiB tables,
# the rollout's request pattern (load + CAS + store, claim rate 0.7).
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04v; mkdir -p $OUT
for cap in 29 30; do
  for draw in 1 2 3 4 5 6; do
    for mib in 2 8 32 64; do
      timeout -k 10 200 tools/variants/exp_requests $cap 20 64 $cap "load+cas+store" 4 717 $mib 2>>$OUT/err.log | python3 -c "
import json, sys; d = json.load(sys.stdin); r = {x['requests']: x['us'] for x in d['rows']}
print('cap 2^$cap draw $draw chunk ${mib} MiB:', r.get('load+cas+store'))" | tee -a $OUT/chunk_size_draws.txt
    done
  done
done
