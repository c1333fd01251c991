#!/usr/bin/env bash
# round 6: counter passes on the FROZEN table's workload (bench.py --prefill-load 0.502: the table pre-filled to just above
# freeze_load, the key set closed at the first launch) -- what a step of a long run moves through the fabric
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r06fzo; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --prefill-load 0.502 --no-companions --cpu-seconds 0 > $OUT/bench_frozen_k20.json 2> $OUT/bench_frozen_k20.err
rc=$?; cut -c1-300 $OUT/bench_frozen_k20.json; [ $rc -eq 0 ] || { tail -n 5 $OUT/bench_frozen_k20.err; exit $rc; }
bash tools/pmc_session.sh r06fz --steps 20 --warmup 5 --prefill-load 0.502 2>&1 | tail -n 3 | cut -c1-300
grep TRAFFIC_JSON gpurun_out/r06fz/pmc/summary.txt | sed 's/^TRAFFIC_JSON //' | python3 -c "import sys,json; print(json.dumps(json.loads(sys.stdin.read()), indent=1))" > $OUT/pmc_traffic_frozen_k20.json
grep -v TRAFFIC_JSON gpurun_out/r06fz/pmc/summary.txt | grep -v "k_table_probe\|k_table_export" > $OUT/pmc_summary_frozen_k20.txt
exit 0
