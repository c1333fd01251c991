#!/usr/bin/env bash
# Round 3, session 3: bucketised probing + 4-call API rework (two board buffers, row cache, masked
# reset): tests, the 4-call timing + per-kernel trace, the driver's bench command.
set -u
TAG=${1:-r03c}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > "$OUT/pytest_gpu.log" 2>&1; rc=$?
tail -n 25 "$OUT/pytest_gpu.log"; echo "pytest rc=$rc"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
echo "== 4-call API"
timeout -k 10 300 python3 tools/archive/exp_unfused.py > "$OUT/four_call.jsonl" 2> "$OUT/four_call.err"; echo "rc=$?"; cat "$OUT/four_call.jsonl"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/four_call_prof" -- \
    python3 tools/archive/exp_unfused.py > "$OUT/four_call_prof.jsonl" 2> "$OUT/four_call_prof.err"; rc=$?
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
find "$OUT/four_call_prof" -name "*kernel_trace.csv" | while read -r f; do
  python3 tools/trace_by_grid.py "$f" "$OUT/four_call_by_grid.txt"
done
find "$OUT/four_call_prof" -name "*kernel_stats.csv" | while read -r f; do cp "$f" "$OUT/four_call_kernel_stats.csv"; done
rm -rf "$OUT/four_call_prof"
echo "== bench, driver command"
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_k20.json" 2> "$OUT/bench_k20.err"; echo "rc=$?"
cat "$OUT/bench_k20.json"; tail -n 3 "$OUT/bench_k20.err"
