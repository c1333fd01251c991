#!/usr/bin/env bash
# round 6: line summaries (Q2048_FLAG_LINE_SUMMARY) -- parity first, then the frozen workload with and without them
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r06s; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 700 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "closed_key_set or freezes or full_table or checkpoint or line_summaries or full_size_1m_lanes_closed" > $OUT/pytest.txt 2>&1
rc=$?; tail -n 3 $OUT/pytest.txt | cut -c1-200; [ $rc -eq 0 ] || { grep -E "Error|assert" $OUT/pytest.txt | head -n 20; exit $rc; }
run() { timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-companions --cpu-seconds 0 "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']; s = d['stats']
print(json.dumps({'args': sys.argv[1], 'us_per_step': round(d['ms_per_step'] * 1e3, 2), 'frozen': c.get('frozen'), 'drops_per_step': s['drops'] / max(1, 20 * 1048576), 'valid': round(s['valid_move_frac'], 3)}))" "$*"; }
run --prefill-load 0.502 | tee -a $OUT/summary.jsonl
run --prefill-load 0.502 --eps 0.01 | tee -a $OUT/summary.jsonl
run --prefill-load 0.502 --steps 256 --warmup 64 | tee -a $OUT/summary.jsonl
run --prefill-load 0.502 --board-size 5 | tee -a $OUT/summary.jsonl
run | tee -a $OUT/summary.jsonl
echo "== train.py 1048576 x 300 (crosses the freeze)"
timeout -k 10 600 python3 train.py --num-envs 1048576 --episodes 300 2>&1 | grep -v amdgpu.ids | grep -v "^epoch [0-9]*[1-9]/" | grep -v "^epoch [0-9]*[1234678]0/" | cut -c1-220 | tee $OUT/train_1048576x300.log
exit 0
