#!/usr/bin/env bash
# The rollout's request pattern (load + CAS + store, claim rate 0.7) on 8 / 16 / 32 GiB tables mapped from
# chunks of 2 ... 64 MiB: which chunk sizes keep a table fast (growing tables are mapped from <= 8192 chunks).
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04v; mkdir -p $OUT
for cap in 28 29 30; do
  for mib in 2 4 8 16 64; do
    timeout -k 10 200 tools/variants/exp_requests $cap 20 64 $cap "load+cas+store" 4 717 $mib > $OUT/req_cap${cap}_chunk${mib}.json 2>>$OUT/err.log || { tail -2 $OUT/err.log; continue; }
    python3 -c "
import json; d = json.load(open('$OUT/req_cap${cap}_chunk${mib}.json')); r = {x['requests']: x['us'] for x in d['rows']}
print('cap 2^$cap chunk ${mib} MiB:', r.get('load+cas+store'), r.get('load+cas+store32'))" | tee -a $OUT/chunk_size.txt
  done
done
