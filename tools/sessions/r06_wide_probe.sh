#!/usr/bin/env bash
# round 6: the closed key set's line-wide probe -- parity first, then the frozen workload (4x4, 5x5, eps 0.95 and 0.01)
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r06w; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "closed_key_set or freezes or full_table or checkpoint" > $OUT/pytest.txt 2>&1
rc=$?; tail -n 3 $OUT/pytest.txt | cut -c1-200; [ $rc -eq 0 ] || exit $rc
run() { timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-companions --cpu-seconds 0 "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']; s = d['stats']
print(json.dumps({'args': sys.argv[1], 'us_per_step': round(d['ms_per_step'] * 1e3, 2), 'frozen': c.get('frozen'), 'drops_per_step': s['drops'] / max(1, 20 * 1048576), 'valid': round(s['valid_move_frac'], 3)}))" "$*"; }
run --prefill-load 0.502 | tee -a $OUT/wide_probe.jsonl
run --prefill-load 0.502 --eps 0.01 | tee -a $OUT/wide_probe.jsonl
run --prefill-load 0.502 --board-size 5 | tee -a $OUT/wide_probe.jsonl
run --prefill-load 0.502 --steps 256 --warmup 64 | tee -a $OUT/wide_probe.jsonl
run | tee -a $OUT/wide_probe.jsonl
exit 0
