#!/usr/bin/env bash
# Round 3 evidence, part A: the whole GPU suite, smoke, bench (default and the driver's command),
# rocprofv3 kernel trace + stats of the default command.
set -u
TAG=${1:-r03h}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
bad() { [ "$1" -eq 124 ] || [ "$1" -eq 137 ]; }
timeout -k 10 1000 python -m pytest tests -m gpu -q -rA > "$OUT/pytest_gpu.log" 2>&1; rc=$?
tail -n 6 "$OUT/pytest_gpu.log" | cut -c1-300; echo "pytest rc=$rc"; bad $rc && exit 1
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; rc=$?; tail -n 2 "$OUT/smoke.log"; bad $rc && exit 1
echo "== bench, driver command"
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_k20.json" 2> "$OUT/bench_k20.err"; rc=$?; cut -c1-300 "$OUT/bench_k20.json"; bad $rc && exit 1
echo "== bench default"
timeout -k 10 500 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; rc=$?; cut -c1-300 "$OUT/bench.json"; bad $rc && exit 1
echo "== rocprofv3 --kernel-trace --stats of the default command (no companions: every k_fused_rollout<4, 0, 0>"
echo "   dispatch is a 64-step learning launch of the measured configuration)"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 bench.py --no-companions --cpu-seconds 0 > "$OUT/prof_bench.json" 2> "$OUT/prof.err"; rc=$?; echo "rc=$rc"; bad $rc && exit 1
find "$OUT/prof" -name "*kernel_stats.csv" | head -n 1 | while read -r f; do cp "$f" "$OUT/kernel_stats.csv"; cut -d, -f1-4,8 "$f" | head -n 8; done
find "$OUT/prof" -name "*kernel_trace.csv" | head -n 1 | while read -r f; do head -n 1 "$f" > "$OUT/kernel_trace_fused.csv"; grep k_fused_rollout "$f" >> "$OUT/kernel_trace_fused.csv"; python3 tools/trace_by_grid.py "$f" "$OUT/kernel_by_grid.txt" > /dev/null; done
rm -rf "$OUT/prof"
echo "== rocprofv3 of the driver's command"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof20" -- python3 bench.py --steps 20 --warmup 5 --no-companions --cpu-seconds 0 > "$OUT/prof_bench_k20.json" 2> "$OUT/prof20.err"; rc=$?; echo "rc=$rc"; bad $rc && exit 1
find "$OUT/prof20" -name "*kernel_stats.csv" | head -n 1 | while read -r f; do cp "$f" "$OUT/kernel_stats_k20.csv"; cut -d, -f1-4,8 "$f" | head -n 6; done
rm -rf "$OUT/prof20"
