#!/usr/bin/env bash
# Round profile: bench (default command), rocprofv3 kernel trace + stats of the same command,
# bench variants, ablations, train.py runs. Outputs -> gpurun_out/<tag>/. (PMC passes: pmc_session.sh)
set -u
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
bad() { [ "$1" -eq 124 ] || [ "$1" -eq 137 ]; }

echo "== bench default"
timeout -k 10 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; rc=$?; cat "$OUT/bench.json"; bad $rc && exit 1
echo "== rocprofv3 --kernel-trace --stats of the same command (without the companion runs, so that every"
echo "   k_fused_rollout<4, 0> dispatch is a 64-step learning launch of the measured configuration: 1 warm-up +"
echo "   5 x 4 timed; the input synthesis runs as k_fused_rollout<4, 4>, the learner-less instantiation)"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 bench.py --no-companions --cpu-seconds 0 > "$OUT/prof_bench.json" 2> "$OUT/prof.err"; rc=$?; echo "rc=$rc"; bad $rc && exit 1
find "$OUT/prof" -name "*kernel_stats.csv" | head -n 1 | while read -r f; do cut -d, -f1-4,8 "$f" | head -n 8; done
echo "== driver-style short run"
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_k20.json" 2> "$OUT/bench_k20.err"; rc=$?; cut -c1-400 "$OUT/bench_k20.json"; bad $rc && exit 1
echo "== bench variants"
for extra in "--strict-td" "--eps 0.01 --strict-td" "--steps-per-launch 1 --steps 64" "--steps-per-launch 16" "--boards-per-gpu 65536 --steps 512 --cap-log2 30" "--board-size 5" "--board-size 5 --eps 0.01" "--agent row-tuple --boards-per-gpu 65536 --steps 512" "--agent row-tuple" "--cap-log2 30 --placement plain" "--cap-log2 30 --placement 4"; do
  echo "-- $extra"
  timeout -k 10 300 python bench.py --cpu-seconds 0 --no-companions $extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'args':sys.argv[1]}|{k:d[k] for k in ('value','ms_per_step','region_ms')}|{'frac':d['roofline']['frac'],'ins':d['stats']['inserts_per_step'],'retries':d['stats']['cas_retries'],'drops':d['stats']['drops'],'episodes':d['stats']['episodes']}))" "$extra" | tee -a "$OUT/variants.jsonl"; rc=${PIPESTATUS[0]}; bad $rc && exit 1
done
echo "== ablation"
timeout -k 10 600 python tools/archive/exp_ablate.py 2> /dev/null | tee "$OUT/ablate.jsonl"
echo "== batch sweep"
timeout -k 10 600 python tools/archive/exp_bsweep.py 2> /dev/null | tee "$OUT/bsweep.jsonl"
echo "== train.py"
timeout -k 10 300 python train.py --num-envs 1 --episodes 3 --log "$OUT/train_single.csv" --summary "$OUT/train_single_summary.csv" 2>&1 | tail -n 2
timeout -k 10 300 python train.py --num-envs 4096 --episodes 50 --episode-log "$OUT/train_220k_episodes.csv" --summary "$OUT/train_220k_summary.csv" --log "$OUT/train_220k_epochs.csv" 2>&1 | tail -n 2
cat "$OUT/train_220k_summary.csv"; rm -f "$OUT/train_220k_episodes.csv"
timeout -k 10 300 python train.py --num-envs 65536 --episodes 40 --log "$OUT/train_65536x40.csv" 2>&1 | tail -n 2
exit 0
