#!/usr/bin/env bash
# Round profile: bench (default command), rocprofv3 kernel trace + stats of the same command,
# PMC passes (one counter set per pass), extra bench configurations. Outputs -> gpurun_out/<tag>/.
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p "$OUT/pmc"
export TMPDIR=/tmp
bad() { [ "$1" -eq 124 ] || [ "$1" -eq 137 ]; }

echo "== bench default"
timeout -k 10 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; rc=$?; cat "$OUT/bench.json"; bad $rc && exit 1
echo "== rocprofv3 --kernel-trace --stats of the same command"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 bench.py > "$OUT/prof_bench.json" 2> "$OUT/prof.err"; rc=$?; echo "rc=$rc"; bad $rc && exit 1
find "$OUT/prof" -name "*kernel_stats.csv" | head -n 1 | while read -r f; do head -n 6 "$f"; done
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_EA0_ATOMIC_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU"; do
  i=$((i+1))
  echo "== pmc pass $i: $set"
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc/pass$i" -- python3 bench.py --cpu-seconds 0 > "$OUT/pmc/pass$i.json" 2> "$OUT/pmc/pass$i.err"; rc=$?
  echo "rc=$rc"; bad $rc && exit 1
done
python3 tools/pmc_summary.py "$OUT/pmc" 1048576 64 | tee "$OUT/pmc/summary.txt"
echo "== bench variants"
for extra in "--strict-td" "--eps 0.01" "--eps 0.01 --strict-td" "--steps-per-launch 1 --steps 64" "--steps-per-launch 16" "--boards-per-gpu 65536 --steps 512" "--board-size 5" "--agent row-tuple --boards-per-gpu 65536 --steps 512" "--agent row-tuple"; do
  echo "-- $extra"
  timeout -k 10 300 python bench.py --cpu-seconds 0 $extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'args':sys.argv[1]}|{k:d[k] for k in ('value','ms_per_step')}|{'frac':d['roofline']['frac'],'retries':d['stats']['cas_retries'],'drops':d['stats']['drops']}))" "$extra" | tee -a "$OUT/variants.jsonl"; rc=${PIPESTATUS[0]}; bad $rc && exit 1
done
echo "== ablation + env-only kernel"
timeout -k 10 600 python tools/exp_ablate.py 2> /dev/null | tee "$OUT/ablate.jsonl"
timeout -k 10 300 python tools/exp_variants.py 2> /dev/null | head -n 2 | tee "$OUT/env_only.jsonl"
echo "== train.py smoke"
timeout -k 10 300 python train.py --num-envs 1 --episodes 3 --log "$OUT/train_single.csv" --summary "$OUT/train_single_summary.csv" 2>&1 | tail -n 3
cat "$OUT/train_single_summary.csv"
timeout -k 10 300 python train.py --num-envs 16384 --board-size 5 --episodes 2 --steps-per-launch 32 --report-every 8 --log "$OUT/train_5x5.csv" --episode-log "$OUT/train_5x5_episodes.csv" --summary "$OUT/train_5x5_summary.csv" 2>&1 | tail -n 2
cat "$OUT/train_5x5_summary.csv"
timeout -k 10 300 python train.py --num-envs 65536 --episodes 3 --steps-per-launch 32 --report-every 4 --log "$OUT/train_batched.csv" 2>&1 | tail -n 4
head -n 4 "$OUT/train_single.csv" "$OUT/train_batched.csv"
