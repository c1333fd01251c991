#!/usr/bin/env bash
# Round 3, session 6: learning quality by write mode at 1 Mi envs, the oracle's thread scaling on the
# box's host, the one-env adapters, deterministic-mode timing + per-kernel timeline.
set -u
TAG=${1:-r03f}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
echo "== learning A/B at 1 Mi envs"
timeout -k 10 500 python3 tools/exp_learning_ab.py --num-envs 1048576 --episodes 4 --seeds 2 --last 1 \
    --capacity-log2 32 --modes store/64,store/1,det > "$OUT/learning_ab_1m.jsonl" 2> "$OUT/learning_ab_1m.err"; rc=$?
echo "rc=$rc"; cut -c1-700 "$OUT/learning_ab_1m.jsonl"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
echo "== oracle thread scaling (host only)"
timeout -k 10 300 python3 tests/cpu_baseline_threads.py > "$OUT/cpu_mt.jsonl" 2> "$OUT/cpu_mt.err"; echo "rc=$?"; cat "$OUT/cpu_mt.jsonl"
echo "== one-env adapters"
timeout -k 10 200 python3 tools/archive/exp_adapters.py > "$OUT/adapters.json" 2> "$OUT/adapters.err"; echo "rc=$?"; cat "$OUT/adapters.json"
echo "== deterministic mode"
timeout -k 10 300 python3 tools/archive/exp_det.py > "$OUT/det.jsonl" 2> "$OUT/det.err"; echo "rc=$?"; cat "$OUT/det.jsonl"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$OUT/det_prof" -- python3 tools/archive/exp_det.py > "$OUT/det_prof.jsonl" 2> "$OUT/det_prof.err"; rc=$?
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
find "$OUT/det_prof" -name "*kernel_trace.csv" | head -n 1 | while read -r f; do python3 tools/archive/det_timeline.py "$f" | tee "$OUT/det_timeline.txt"; done
rm -rf "$OUT/det_prof"
