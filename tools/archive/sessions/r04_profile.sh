#!/usr/bin/env bash
# Round 4 profile on the final kernel sources: growth phases + train.py on a growing table, the bench lines
# (default command, driver's command), rocprofv3 --kernel-trace --stats of both, a few variants, the other
# entry points (4-call loop, deterministic step), the train.py runs.  PMC passes: tools/pmc_session.sh.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04p; mkdir -p $OUT
export TMPDIR=/tmp
bad() { [ "$1" -eq 124 ] || [ "$1" -eq 137 ]; }
echo "== growth phases (chunk count bounded)"
timeout -k 10 600 python3 tools/archive/exp_grow.py > $OUT/grow.txt 2>&1; rc=$?; tail -n 14 $OUT/grow.txt | cut -c1-260; bad $rc && exit 1
echo "== train.py on a growing table: 262 144 envs x 100 episodes"
timeout -k 10 600 python3 train.py --num-envs 262144 --episodes 100 --log $OUT/train_262144x100.csv 2>&1 | grep -v "^epoch [0-9]*[1-9]/" | tail -n 20 | tee $OUT/train_262144x100.log
echo "== bench, driver's command"
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err; rc=$?; cut -c1-500 $OUT/bench_k20.json; bad $rc && exit 1
echo "== rocprofv3 --kernel-trace --stats, driver's command"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_k20 -- python3 bench.py --steps 20 --warmup 5 --no-companions --cpu-seconds 0 > $OUT/prof_bench_k20.json 2> $OUT/prof_k20.err; rc=$?; echo "rc=$rc"; bad $rc && exit 1
find $OUT/prof_k20 -name "*kernel_stats.csv" | head -n 1 | while read -r f; do cp "$f" $OUT/kernel_stats_k20.csv; cut -d, -f1-4,8 "$f" | head -n 6; done
find $OUT/prof_k20 -name "*kernel_trace.csv" | head -n 1 | while read -r f; do grep -c "k_fused_rollout" "$f"; python3 tools/trace_by_grid.py "$f" > $OUT/kernel_by_grid_k20.txt 2>/dev/null || true; done
rm -rf $OUT/prof_k20
echo "== bench, default command"
timeout -k 10 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; rc=$?; cut -c1-500 $OUT/bench.json; bad $rc && exit 1
echo "== rocprofv3 --kernel-trace --stats, default command"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --no-companions --cpu-seconds 0 > $OUT/prof_bench.json 2> $OUT/prof.err; rc=$?; echo "rc=$rc"; bad $rc && exit 1
find $OUT/prof -name "*kernel_stats.csv" | head -n 1 | while read -r f; do cp "$f" $OUT/kernel_stats.csv; cut -d, -f1-4,8 "$f" | head -n 6; done
rm -rf $OUT/prof
echo "== variants"
for extra in "--board-size 5" "--boards-per-gpu 65536 --steps 512 --cap-log2 30" "--agent row-tuple --boards-per-gpu 65536 --steps 512" "--eps 0.01" "--strict-td"; do
  timeout -k 10 300 python3 bench.py --cpu-seconds 0 --no-companions $extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'args':sys.argv[1]}|{k:d[k] for k in ('value','ms_per_step','region_ms')}|{'frac':d['roofline']['frac'],'ins':d['stats']['inserts_per_step'],'drops':d['stats']['drops'],'episodes':d['stats']['episodes']}))" "$extra" | tee -a $OUT/variants.jsonl; rc=${PIPESTATUS[0]}; bad $rc && exit 1
done
echo "== the other entry points"
timeout -k 10 300 python3 tools/archive/exp_unfused.py 2>/dev/null | tee $OUT/four_call.jsonl
timeout -k 10 300 python3 tools/archive/exp_det.py 2>/dev/null | tee $OUT/deterministic_mode.jsonl
timeout -k 10 300 python3 tools/archive/exp_adapters.py 2>/dev/null | tee $OUT/adapters.json
echo "== train.py, the other sizes"
timeout -k 10 300 python3 train.py --num-envs 4096 --episodes 50 --episode-log $OUT/train_220k_episodes.csv --summary $OUT/train_220k_summary.csv --log $OUT/train_220k_epochs.csv 2>&1 | tail -n 2
cat $OUT/train_220k_summary.csv; rm -f $OUT/train_220k_episodes.csv
timeout -k 10 300 python3 train.py --num-envs 65536 --episodes 40 --log $OUT/train_65536x40.csv 2>&1 | tail -n 2
exit 0
