#!/usr/bin/env bash
# One gpurun session: GPU parity tests, smoke, bench, rocprof kernel trace. Logs -> gpurun_out/.
# Usage (from the repo root on the GPU box):  bash tools/gpu_session.sh [tag]
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }

echo "== pytest -m gpu" | tee "$OUT/session.log"
timeout -k 10 900 python -m pytest tests -m gpu -q -rA -s > "$OUT/pytest_gpu.log" 2>&1; rc=$?
tail -n 40 "$OUT/pytest_gpu.log" | tee -a "$OUT/session.log"
echo "pytest rc=$rc" | tee -a "$OUT/session.log"
ok $rc || exit 1

echo "== smoke" | tee -a "$OUT/session.log"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; rc=$?
cat "$OUT/smoke.log" | tee -a "$OUT/session.log"
ok $rc || exit 1

echo "== bench (short)" | tee -a "$OUT/session.log"
timeout -k 10 600 python bench.py --steps 64 --warmup 32 --cpu-seconds 6 > "$OUT/bench_short.json" 2> "$OUT/bench_short.err"; rc=$?
cat "$OUT/bench_short.json" | tee -a "$OUT/session.log"; tail -n 5 "$OUT/bench_short.err"
ok $rc || exit 1
[ $rc -eq 0 ] || exit 1

echo "== bench (default)" | tee -a "$OUT/session.log"
timeout -k 10 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; rc=$?
cat "$OUT/bench.json" | tee -a "$OUT/session.log"; tail -n 5 "$OUT/bench.err"
ok $rc || exit 1

echo "== rocprofv3 kernel trace" | tee -a "$OUT/session.log"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- \
    python3 bench.py --steps 64 --warmup 32 --cpu-seconds 0 > "$OUT/prof_bench.json" 2> "$OUT/prof.err"; rc=$?
echo "rocprof rc=$rc" | tee -a "$OUT/session.log"
find "$OUT/prof" -name "*kernel_stats.csv" | head -n 3 | while read -r f; do echo "-- $f"; head -n 12 "$f"; done | tee -a "$OUT/session.log"
exit 0
