#!/usr/bin/env bash
# strict-td: does bounding the CAS retries remove the low-epsilon storm?
set -u
OUT=gpurun_out/${1:-r02h}; mkdir -p "$OUT"; export TMPDIR=/tmp
for bits in 0 0x10000 0x20000 0x40000 0x100000; do
  for eps in 0.95 0.01; do
    timeout -k 10 300 python bench.py --cpu-seconds 0 --no-companions --strict-td --repeats 3 --eps $eps --experiment-bits $bits 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'max_cas_bits': sys.argv[1], 'eps': d['config']['epsilon'], 'ms_per_step': d['ms_per_step'], 'frac': d['roofline']['frac'], 'region_ms': d['region_ms'], 'retries': d['stats']['cas_retries']}))" $bits | tee -a "$OUT/strict.jsonl"
  done
done
