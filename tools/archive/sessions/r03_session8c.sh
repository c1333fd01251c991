#!/usr/bin/env bash
# Round 3 evidence, part C: bench variants, ablations, sweeps, train.py runs, 4-call API trace.
set -u
TAG=${1:-r03j}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
bad() { [ "$1" -eq 124 ] || [ "$1" -eq 137 ]; }
echo "== bench variants"
for extra in "--strict-td" "--eps 0.01 --strict-td" "--steps-per-launch 1 --steps 64" "--steps-per-launch 16" "--boards-per-gpu 65536 --steps 512 --cap-log2 30" "--board-size 5" "--board-size 5 --eps 0.01" "--agent row-tuple --boards-per-gpu 65536 --steps 512" "--agent row-tuple" "--cap-log2 30 --placement plain" "--cap-log2 30 --placement chunks" "--cap-log2 30 --placement 4"; do
  echo "-- $extra"
  timeout -k 10 300 python bench.py --cpu-seconds 0 --no-companions $extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'args':sys.argv[1]}|{k:d[k] for k in ('value','ms_per_step','region_ms')}|{'frac':d['roofline']['frac'],'ins':d['stats']['inserts_per_step'],'retries':d['stats']['cas_retries'],'drops':d['stats']['drops'],'episodes':d['stats']['episodes'],'placement':d['config']['table_placement']}))" "$extra" | tee -a "$OUT/variants.jsonl" | cut -c1-300; rc=${PIPESTATUS[0]}; bad $rc && exit 1
done
echo "== ablation"
timeout -k 10 500 python tools/archive/exp_ablate.py 2> /dev/null | tee "$OUT/ablate.jsonl"
echo "== batch sweep"
timeout -k 10 400 python tools/archive/exp_bsweep.py 2> /dev/null | tee "$OUT/bsweep.jsonl"
echo "== 4-call API"
timeout -k 10 200 python3 tools/archive/exp_unfused.py > "$OUT/four_call.jsonl" 2> /dev/null; cat "$OUT/four_call.jsonl"
echo "== train.py"
timeout -k 10 300 python train.py --num-envs 4096 --episodes 50 --episode-log "$OUT/train_220k_episodes.csv" --summary "$OUT/train_220k_summary.csv" --log "$OUT/train_220k_epochs.csv" 2>&1 | tail -n 2
cat "$OUT/train_220k_summary.csv"; rm -f "$OUT/train_220k_episodes.csv"
timeout -k 10 300 python train.py --num-envs 65536 --episodes 40 --log "$OUT/train_65536x40.csv" 2>&1 | tail -n 2
timeout -k 10 400 python train.py --num-envs 262144 --episodes 100 --capacity-log2 32 --log "$OUT/train_262144x100.csv" 2>&1 | tail -n 2
exit 0
