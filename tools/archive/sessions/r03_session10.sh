#!/usr/bin/env bash
# Round 3, final evidence after the statistics-atomics change: session 9 (suite, smoke, benches, kernel traces,
# PMC passes) + the 4-call API's per-kernel trace + the deterministic step's timing and timeline.
set -u
TAG=${1:-r03n}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
bash tools/archive/sessions/r03_session9.sh $TAG || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
echo "== 4-call API"
timeout -k 10 200 python3 tools/archive/exp_unfused.py > "$OUT/four_call.jsonl" 2> /dev/null; cat "$OUT/four_call.jsonl"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/fc" -- python3 tools/archive/exp_unfused.py > /dev/null 2>&1
find "$OUT/fc" -name "*kernel_trace.csv" | head -n 1 | while read -r f; do python3 tools/trace_by_grid.py "$f" "$OUT/four_call_by_grid.txt" | head -n 14; done
find "$OUT/fc" -name "*kernel_stats.csv" | head -n 1 | while read -r f; do cp "$f" "$OUT/four_call_kernel_stats.csv"; done
rm -rf "$OUT/fc"
echo "== deterministic mode"
timeout -k 10 300 python3 tools/archive/exp_det.py > "$OUT/det.jsonl" 2> /dev/null; cat "$OUT/det.jsonl"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$OUT/dp" -- python3 tools/archive/exp_det.py > /dev/null 2>&1
find "$OUT/dp" -name "*kernel_trace.csv" | head -n 1 | while read -r f; do python3 tools/archive/det_timeline.py "$f" | tee "$OUT/det_timeline.txt"; done
rm -rf "$OUT/dp"
