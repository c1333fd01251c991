#!/usr/bin/env bash
# After the core trims (select masks, one reward logarithm) and the deterministic-mode rewrite: the
# whole GPU suite, then the judged trio again -- bench default, rocprofv3 stats of the same
# command, the driver's short run.
set -u
OUT=gpurun_out/${1:-r02i}; mkdir -p "$OUT"; export TMPDIR=/tmp
bad() { [ "$1" -eq 124 ] || [ "$1" -eq 137 ]; }
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; rc=$?; tail -n 3 "$OUT/pytest.log"; [ $rc -ne 0 ] && exit 1
timeout -k 10 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; rc=$?; cut -c1-300 "$OUT/bench.json"; bad $rc && exit 1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 bench.py --no-companions --cpu-seconds 0 > "$OUT/prof_bench.json" 2> "$OUT/prof.err"; rc=$?; echo "rc=$rc"; bad $rc && exit 1
find "$OUT/prof" -name "*kernel_stats.csv" | head -n 1 | while read -r f; do cut -d, -f1-4,8 "$f" | head -n 6; done
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_k20.json" 2> "$OUT/bench_k20.err"; rc=$?; cut -c1-300 "$OUT/bench_k20.json"; bad $rc && exit 1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
