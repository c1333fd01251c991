#!/usr/bin/env bash
# Round-2 GPU session 2: full parity suite, scan timing, deferred-write A/B, PMC passes.
set -u
TAG=${1:-r02b}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
bad() { [ "$1" -eq 124 ] || [ "$1" -eq 137 ]; }
echo "== pytest -m gpu"
timeout -k 10 1100 python -m pytest tests -m gpu -q -rA -s > "$OUT/pytest_gpu.log" 2>&1; rc=$?
grep -E "passed|failed|FAILED|Error" "$OUT/pytest_gpu.log" | tail -n 15; echo "pytest rc=$rc"; bad $rc && exit 1
echo "== table scan timing"
timeout -k 10 300 python tools/archive/exp_export.py > "$OUT/export.jsonl" 2> "$OUT/export.err"; rc=$?
cat "$OUT/export.jsonl"; tail -n 3 "$OUT/export.err"; bad $rc && exit 1
echo "== bench, deferred same-state writes (default) vs immediate (bit 14)"
for bits in 0 0x4000; do
  for eps in 0.95 0.01; do
    timeout -k 10 300 python bench.py --cpu-seconds 0 --no-companions --eps $eps --experiment-bits $bits 2> "$OUT/ab.err" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'bits': sys.argv[1], 'eps': d['config']['epsilon'], 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'frac': d['roofline']['frac'], 'region_ms': d['region_ms'], 'ins': d['stats']['inserts_per_step'], 'valid': d['stats']['valid_move_frac']}))" $bits | tee -a "$OUT/ab_defer.jsonl"; rc=${PIPESTATUS[0]}; bad $rc && exit 1
  done
done
echo "== PMC passes"
bash tools/pmc_session.sh $TAG
exit 0
