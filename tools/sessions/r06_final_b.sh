#!/usr/bin/env bash
# round 6, closing session B: both bench lines (with `roofline.traffic` from session A's passes, committed under profiles/),
# rocprofv3 --kernel-trace --stats of both commands, the adapters and the 4-call loop
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r06z; mkdir -p $OUT
export TMPDIR=/tmp
bad() { [ "$1" -eq 124 ] || [ "$1" -eq 137 ]; }
echo "== bench, driver's command"
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err; rc=$?; cut -c1-200 $OUT/bench_k20.json; bad $rc && exit 1
echo "== bench, default command"
timeout -k 10 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; rc=$?; cut -c1-200 $OUT/bench.json; bad $rc && exit 1
echo "== rocprofv3 --kernel-trace --stats, driver's command"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_k20 -- python3 bench.py --steps 20 --warmup 5 --no-companions --cpu-seconds 0 > $OUT/bench_profiled_run_k20.json 2> $OUT/prof_k20.err; rc=$?; echo "rc=$rc"; bad $rc && exit 1
find $OUT/prof_k20 -name "*kernel_stats.csv" | head -n 1 | while read -r f; do cp "$f" $OUT/kernel_stats_k20.csv; done
find $OUT/prof_k20 -name "*kernel_trace.csv" | head -n 1 | while read -r f; do python3 tools/trace_by_grid.py "$f" > $OUT/kernel_by_grid_k20.txt 2>/dev/null || true; done
rm -rf $OUT/prof_k20; head -n 6 $OUT/kernel_by_grid_k20.txt
echo "== rocprofv3 --kernel-trace --stats, default command"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --no-companions --cpu-seconds 0 > $OUT/bench_profiled_run.json 2> $OUT/prof.err; rc=$?; echo "rc=$rc"; bad $rc && exit 1
find $OUT/prof -name "*kernel_stats.csv" | head -n 1 | while read -r f; do cp "$f" $OUT/kernel_stats.csv; cut -d, -f1-4 "$f" | head -n 3 | cut -c1-200; done
rm -rf $OUT/prof
timeout -k 10 300 python3 tools/archive/exp_adapters.py 2>/dev/null | tee $OUT/adapters.json
timeout -k 10 300 python3 tools/archive/exp_unfused.py 2>/dev/null | tee $OUT/four_call.jsonl
timeout -k 10 300 python3 tools/archive/exp_det.py 2>/dev/null | tee $OUT/deterministic_mode.jsonl
echo "== fuzz parity (random geometry / batch / splits / 4-call switches / a key set that closes at a random step)"
FUZZ_PRODUCT=1 timeout -k 10 400 python3 tests/fuzz_parity.py 601 12 2>/dev/null | tail -n 13 | cut -c1-260 | tee $OUT/fuzz_product.txt
timeout -k 10 400 python3 tests/fuzz_parity.py 602 12 2>/dev/null | tail -n 13 | cut -c1-260 | tee $OUT/fuzz_experiments.txt
exit 0
