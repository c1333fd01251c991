#!/usr/bin/env bash
# Round-2 GPU session 1: parity tests, smoke, request micro-benchmarks (full-line stores, occupancy
# bitmap), table-scan timing, bench (default + 2-rank gloo rehearsal). Logs -> gpurun_out/<tag>/.
set -u
TAG=${1:-r02a}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
bad() { [ "$1" -eq 124 ] || [ "$1" -eq 137 ]; }

echo "== pytest -m gpu"
timeout -k 10 1000 python -m pytest tests -m gpu -q -x -rA -s > "$OUT/pytest_gpu.log" 2>&1; rc=$?
tail -n 25 "$OUT/pytest_gpu.log"; echo "pytest rc=$rc"; bad $rc && exit 1

echo "== smoke"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; rc=$?
tail -n 3 "$OUT/smoke.log"; bad $rc && exit 1

echo "== request micro-benchmark, 2^32 slots"
timeout -k 10 300 tools/variants/exp_requests 32 20 32 > "$OUT/requests_cap32.json" 2> "$OUT/requests_cap32.err"; rc=$?
cat "$OUT/requests_cap32.json"; bad $rc && exit 1
echo "== request micro-benchmark, 2^30 slots"
timeout -k 10 300 tools/variants/exp_requests 30 20 32 > "$OUT/requests_cap30.json" 2> "$OUT/requests_cap30.err"; rc=$?
cat "$OUT/requests_cap30.json"; bad $rc && exit 1

echo "== table scan (len(q_table)) timing"
timeout -k 10 300 python tools/archive/exp_export.py > "$OUT/export.jsonl" 2> "$OUT/export.err"; rc=$?
cat "$OUT/export.jsonl"; tail -n 3 "$OUT/export.err"; bad $rc && exit 1

echo "== bench default"
timeout -k 10 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; rc=$?
cat "$OUT/bench.json"; tail -n 5 "$OUT/bench.err"; echo "bench rc=$rc"; bad $rc && exit 1

echo "== bench driver-style short run"
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --no-companions > "$OUT/bench_k20.json" 2> "$OUT/bench_k20.err"; rc=$?
cat "$OUT/bench_k20.json"; tail -n 3 "$OUT/bench_k20.err"; bad $rc && exit 1

echo "== bench --gpus 2 rehearsal (gloo, both ranks on the one GPU)"
Q2048_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 32 --warmup 8 --boards-per-gpu 262144 --cap-log2 28 --cpu-seconds 0 > "$OUT/bench_2rank_gloo.json" 2> "$OUT/bench_2rank_gloo.err"; rc=$?
cat "$OUT/bench_2rank_gloo.json"; tail -n 5 "$OUT/bench_2rank_gloo.err"; echo "2-rank rc=$rc"
exit 0
