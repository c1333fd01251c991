cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r06o
run() { timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-companions --cpu-seconds 0 "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']; s = d['stats']
print(json.dumps({'args': sys.argv[1], 'us_per_step': round(d['ms_per_step'] * 1e3, 2), 'frozen': c.get('frozen'), 'drops_per_step': s['drops'] / max(1, 20 * 1048576)}))" "$*"; }
run --prefill-load 0.502 --experiment-bits 0x100000 | tee -a gpurun_out/r06o/one_request.jsonl
run --prefill-load 0.502 --experiment-bits 0x8000 | tee -a gpurun_out/r06o/one_request.jsonl
run --prefill-load 0.502 --experiment-bits 0x8000 --eps 0.01 | tee -a gpurun_out/r06o/one_request.jsonl
run --prefill-load 0.502 --experiment-bits 0x2000 | tee -a gpurun_out/r06o/one_request.jsonl
