#!/usr/bin/env bash
# Round 3, session 7: tables mapped from 2 MiB chunks (q2048_table_alloc) -- tests, then the driver's
# bench command at 2^28 / 2^30 / 2^32 slots; then session 6's measurements.
set -u
TAG=${1:-r03g}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "placement or probing or resume or row_cache or step_to or spanning or probe" > "$OUT/pytest_sel.log" 2>&1; rc=$?
tail -n 12 "$OUT/pytest_sel.log" | cut -c1-300; echo "pytest rc=$rc"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
for cap in 28 30 32; do
  for pl in auto plain; do
    echo "== bench --steps 20 --warmup 5 --cap-log2 $cap --placement $pl"
    timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cap-log2 $cap --placement $pl --no-companions --cpu-seconds 0 > "$OUT/bench_k20_cap${cap}_$pl.json" 2> "$OUT/bench_k20_cap${cap}_$pl.err"; rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
    python3 -c "import json;d=json.load(open('$OUT/bench_k20_cap${cap}_$pl.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['stats']['table_load_factor'],d['region_ms'],d['config']['table_placement'])"
  done
done
echo "== 2^28 chunks, 64-step launches to load 0.4"
timeout -k 10 300 python bench.py --cap-log2 28 --steps 64 --warmup 16 --repeats 2 --no-companions --cpu-seconds 0 > "$OUT/bench_cap28_s64.json" 2> "$OUT/bench_cap28_s64.err"
python3 -c "import json;d=json.load(open('$OUT/bench_cap28_s64.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['stats']['table_load_factor'],d['region_ms'],d['config']['table_placement'])"
bash tools/archive/sessions/r03_session6.sh $TAG
