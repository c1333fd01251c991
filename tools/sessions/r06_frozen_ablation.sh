#!/usr/bin/env bash
# round 6: where does the frozen step's time go?  The frozen workload (bench.py --prefill-load 0.502) on the measurement
# build with the ablation bits: 13 = no probe of the next state (arithmetic of the frozen path alone)
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r06ab; mkdir -p $OUT
export TMPDIR=/tmp
run() { timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-companions --cpu-seconds 0 "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']; s = d['stats']
print(json.dumps({'args': sys.argv[1], 'us_per_step': round(d['ms_per_step'] * 1e3, 2), 'frozen': c.get('frozen'), 'bits': c['experiment_bits'], 'drops_per_step': s['drops'] / max(1, 20 * 1048576), 'inserts_per_step': round(s['inserts_per_step'], 3), 'valid': round(s['valid_move_frac'], 3)}))" "$*"; }
run --prefill-load 0.502 | tee -a $OUT/frozen_ablation.jsonl
run --prefill-load 0.502 --experiment-bits 0x100000 | tee -a $OUT/frozen_ablation.jsonl
run --prefill-load 0.502 --experiment-bits 0x2000 | tee -a $OUT/frozen_ablation.jsonl
run --experiment-bits 0x100000 | tee -a $OUT/frozen_ablation.jsonl
run --experiment-bits 0x3000 | tee -a $OUT/frozen_ablation.jsonl
run --prefill-load 0.502 --eps 0.01 | tee -a $OUT/frozen_ablation.jsonl
run --prefill-load 0.502 --eps 0.01 --experiment-bits 0x2000 | tee -a $OUT/frozen_ablation.jsonl
exit 0
