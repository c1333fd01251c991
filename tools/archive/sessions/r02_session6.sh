#!/usr/bin/env bash
# Round-2 GPU session 6: adapters (pinned host-visible memory), deterministic-mode kernel breakdown,
# train.py flags.
set -u
TAG=${1:-r02f}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
bad() { [ "$1" -eq 124 ] || [ "$1" -eq 137 ]; }
echo "== pytest -m gpu"
timeout -k 10 1100 python -m pytest tests -m gpu -q -rA -s > "$OUT/pytest_gpu.log" 2>&1; rc=$?
grep -E "passed|failed|FAILED|Error" "$OUT/pytest_gpu.log" | tail -n 15; echo "pytest rc=$rc"; bad $rc && exit 1
[ $rc -eq 0 ] || { tail -n 80 "$OUT/pytest_gpu.log"; }
echo "== adapters"
timeout -k 10 300 python tools/archive/exp_adapters.py > "$OUT/adapters.json" 2> "$OUT/adapters.err"; rc=$?
cat "$OUT/adapters.json"; tail -n 3 "$OUT/adapters.err"; bad $rc && exit 1
echo "== deterministic mode: kernel breakdown"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_det" -- python3 tools/archive/exp_det.py > "$OUT/det.jsonl" 2> "$OUT/det.err"; rc=$?
cat "$OUT/det.jsonl"; bad $rc && exit 1
find "$OUT/prof_det" -name "*kernel_stats.csv" | head -n 1 | while read -r f; do cut -d, -f1-4,6-8 "$f" | head -n 16; done
echo "== train.py"
timeout -k 10 300 python train.py --num-envs 1 --episodes 3 --reset-shaping-state --log "$OUT/train_single.csv" 2>&1 | tail -n 2
timeout -k 10 300 python train.py --num-envs 16384 --episodes 3 --env-profile nopenalty --steps-per-launch 32 --report-every 8 --log "$OUT/train_nopenalty.csv" 2>&1 | tail -n 2
timeout -k 10 300 python train.py --num-envs 16384 --episodes 3 --reset-shaping-state --deterministic --steps-per-launch 32 --report-every 8 --log "$OUT/train_det.csv" 2>&1 | tail -n 2
exit 0
