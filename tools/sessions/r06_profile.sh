#!/usr/bin/env bash
# Round 6 evidence on the final kernel sources: the whole GPU suite, the bench lines (driver's command, default
# command), rocprofv3 --kernel-trace --stats of both, train.py (growing table; a run that outgrows the largest table and
# freezes; the reference's experiment size with its summary row), the other entry points.
# PMC passes: tools/sessions/r06_pmc.sh (their own session: --pmc never together with another trace domain).
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r06p; mkdir -p $OUT
export TMPDIR=/tmp
bad() { [ "$1" -eq 124 ] || [ "$1" -eq 137 ]; }
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
rc=$?; tail -n 6 $OUT/pytest_gpu.txt | cut -c1-300; echo "pytest rc $rc"; [ $rc -eq 0 ] || exit $rc
T="timeout -k 10 600 python3 train.py"
filt() { grep -v "^epoch [0-9]*[1-9]/" | grep -v amdgpu.ids | tail -n 24; }
echo "== bench, driver's command"
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err; rc=$?; cut -c1-300 $OUT/bench_k20.json; bad $rc && exit 1
echo "== rocprofv3 --kernel-trace --stats, driver's command"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_k20 -- python3 bench.py --steps 20 --warmup 5 --no-companions --cpu-seconds 0 > $OUT/bench_profiled_run_k20.json 2> $OUT/prof_k20.err; rc=$?; echo "rc=$rc"; bad $rc && exit 1
find $OUT/prof_k20 -name "*kernel_stats.csv" | head -n 1 | while read -r f; do cp "$f" $OUT/kernel_stats_k20.csv; cut -d, -f1-4,8 "$f" | head -n 6; done
find $OUT/prof_k20 -name "*kernel_trace.csv" | head -n 1 | while read -r f; do python3 tools/trace_by_grid.py "$f" > $OUT/kernel_by_grid_k20.txt 2>/dev/null || true; done
rm -rf $OUT/prof_k20
echo "== bench, default command"
timeout -k 10 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; rc=$?; cut -c1-300 $OUT/bench.json; bad $rc && exit 1
echo "== rocprofv3 --kernel-trace --stats, default command"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --no-companions --cpu-seconds 0 > $OUT/bench_profiled_run.json 2> $OUT/prof.err; rc=$?; echo "rc=$rc"; bad $rc && exit 1
find $OUT/prof -name "*kernel_stats.csv" | head -n 1 | while read -r f; do cp "$f" $OUT/kernel_stats.csv; cut -d, -f1-4,8 "$f" | head -n 6; done
rm -rf $OUT/prof
echo "== train.py 262144 x 100, default"
$T --num-envs 262144 --episodes 100 --log $OUT/train_262144x100_growing.csv 2>&1 | filt | tee $OUT/train_262144x100_growing.log
echo "== train.py 1048576 x 20, default"
$T --num-envs 1048576 --episodes 20 --log $OUT/train_1048576x20_growing.csv 2>&1 | filt | tee $OUT/train_1048576x20_growing.log
echo "== train.py, the reference's experiment size: 4096 envs x 50 epochs = 200 000 games, with the summary row"
timeout -k 10 300 python3 train.py --num-envs 4096 --episodes 50 --episode-log $OUT/train_220k_episodes.csv --summary $OUT/train_220k_games_summary.csv --log $OUT/train_220k_epochs.csv 2>&1 | tail -n 2
cat $OUT/train_220k_games_summary.csv; rm -f $OUT/train_220k_episodes.csv
echo "== the other entry points"
timeout -k 10 300 python3 tools/archive/exp_unfused.py 2>/dev/null | tee $OUT/four_call.jsonl
timeout -k 10 300 python3 tools/archive/exp_det.py 2>/dev/null | tee $OUT/deterministic_mode.jsonl
timeout -k 10 300 python3 tools/archive/exp_adapters.py 2>/dev/null | tee $OUT/adapters.json
exit 0
