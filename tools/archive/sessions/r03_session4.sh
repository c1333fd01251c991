#!/usr/bin/env bash
# Round 3, session 4: whole GPU suite on the current tree, 5x5 / 4x4 single-step kernels, the 2^28
# long run (bucketised probing at load 0.6), bench (driver command) and its PMC passes.
set -u
TAG=${1:-r03d}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > "$OUT/pytest_gpu.log" 2>&1; rc=$?
tail -n 25 "$OUT/pytest_gpu.log" | cut -c1-400; echo "pytest rc=$rc"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
echo "== env step kernels"
timeout -k 10 300 python3 tools/archive/exp_env_step.py > "$OUT/env_step.jsonl" 2> "$OUT/env_step.err"; echo "rc=$?"; cat "$OUT/env_step.jsonl"
echo "== 2^28 long run"
timeout -k 10 300 python bench.py --cap-log2 28 --no-companions --cpu-seconds 0 > "$OUT/bench_cap28.json" 2> "$OUT/bench_cap28.err"; echo "rc=$?"
python3 -c "import json;d=json.load(open('$OUT/bench_cap28.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['stats']['table_load_factor'],d['region_ms'],d['config']['table_placement'])"
echo "== bench, driver command"
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_k20.json" 2> "$OUT/bench_k20.err"; echo "rc=$?"
python3 -c "import json;d=json.load(open('$OUT/bench_k20.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac']);[print(c['name'],c['value'],c['roofline_frac'],c['table_load_factor'],c['table_placement']) for c in d['companions']]"
echo "== PMC passes, driver command"
bash tools/pmc_session.sh $TAG --steps 20 --warmup 5 > "$OUT/pmc_k20.log" 2>&1; echo "rc=$?"
tail -n 5 "$OUT/pmc_k20.log" | cut -c1-600
# keep the per-pass raw counter CSVs out of the merge (tens of MiB): the summary has what is needed
find "$OUT/pmc" -name "*.csv" -size +2M -delete
