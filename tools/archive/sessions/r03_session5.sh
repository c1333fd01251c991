#!/usr/bin/env bash
# Round 3, session 5: whole GPU suite, the 2^28 run to load 0.45, 8 GiB tables assembled from spread
# physical chunks (VMM), PMC passes of the driver's command.
set -u
TAG=${1:-r03e}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > "$OUT/pytest_gpu.log" 2>&1; rc=$?
tail -n 12 "$OUT/pytest_gpu.log" | cut -c1-300; echo "pytest rc=$rc"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
echo "== 2^28 run (64-step launches, ends near load 0.45)"
timeout -k 10 300 python bench.py --cap-log2 28 --steps 64 --warmup 16 --repeats 2 --no-companions --cpu-seconds 0 > "$OUT/bench_cap28.json" 2> "$OUT/bench_cap28.err"; echo "rc=$?"
python3 -c "import json;d=json.load(open('$OUT/bench_cap28.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['stats']['table_load_factor'],d['region_ms'],d['config']['table_placement'])"
echo "== 8 GiB tables from VMM chunks: consecutive vs spread over 224 GiB"
timeout -k 10 280 tools/variants/exp_vmm 64 224 0 0 8 > "$OUT/vmm_64MiB_8GiB.txt" 2>&1; echo "rc=$?"; cat "$OUT/vmm_64MiB_8GiB.txt"
timeout -k 10 280 tools/variants/exp_vmm 2 224 0 0 8 > "$OUT/vmm_2MiB_8GiB.txt" 2>&1; echo "rc=$?"; cat "$OUT/vmm_2MiB_8GiB.txt"
echo "== PMC passes, driver command"
bash tools/pmc_session.sh $TAG --steps 20 --warmup 5 > "$OUT/pmc_k20.log" 2>&1; echo "rc=$?"
grep TRAFFIC_JSON "$OUT/pmc_k20.log" | cut -c1-400
find "$OUT/pmc" -name "*.csv" -size +2M -delete
