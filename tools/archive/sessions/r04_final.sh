#!/usr/bin/env bash
# Round 4, last session: counters on 5x5, then the bench lines (with roofline.traffic from the committed
# passes) and the rocprofv3 kernel stats of both commands.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04z; mkdir -p $OUT
export TMPDIR=/tmp
bad() { [ "$1" -eq 124 ] || [ "$1" -eq 137 ]; }
echo "== bench, driver's command"
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err; rc=$?; cut -c1-300 $OUT/bench_k20.json; bad $rc && exit 1
echo "== rocprofv3 --kernel-trace --stats, driver's command"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_k20 -- python3 bench.py --steps 20 --warmup 5 --no-companions --cpu-seconds 0 > $OUT/prof_bench_k20.json 2> $OUT/prof_k20.err; rc=$?; echo "rc=$rc"; bad $rc && exit 1
find $OUT/prof_k20 -name "*kernel_stats.csv" | head -n 1 | while read -r f; do cp "$f" $OUT/kernel_stats_k20.csv; cut -d, -f1-4,8 "$f" | head -n 6; done
find $OUT/prof_k20 -name "*kernel_trace.csv" | head -n 1 | while read -r f; do python3 tools/trace_by_grid.py "$f" > $OUT/kernel_by_grid_k20.txt 2>/dev/null || true; done
rm -rf $OUT/prof_k20
echo "== bench, default command"
timeout -k 10 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; rc=$?; cut -c1-300 $OUT/bench.json; bad $rc && exit 1
echo "== rocprofv3 --kernel-trace --stats, default command"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --no-companions --cpu-seconds 0 > $OUT/prof_bench.json 2> $OUT/prof.err; rc=$?; echo "rc=$rc"; bad $rc && exit 1
find $OUT/prof -name "*kernel_stats.csv" | head -n 1 | while read -r f; do cp "$f" $OUT/kernel_stats.csv; cut -d, -f1-4,8 "$f" | head -n 6; done
rm -rf $OUT/prof
echo "== counters, 5x5"
bash tools/pmc_session.sh r04n5 --board-size 5 --cap-log2 32 2>&1 | grep "TRAFFIC_JSON\|rc=" | cut -c1-300
