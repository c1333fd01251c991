#!/usr/bin/env bash
# round 6, closing session: the tests added last, the counter passes of the default command on a 2^32-slot table (the
# table the default bench line runs on), then both bench lines with `roofline.traffic` from this round's passes
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r06z; mkdir -p $OUT
export TMPDIR=/tmp
bad() { [ "$1" -eq 124 ] || [ "$1" -eq 137 ]; }
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
rc=$?; tail -n 4 $OUT/pytest_gpu.txt | cut -c1-300; echo "pytest rc $rc"; [ $rc -eq 0 ] || exit $rc
bash tools/pmc_session.sh r06def --cap-log2 32 2>&1 | tail -n 4 | cut -c1-300
grep TRAFFIC_JSON gpurun_out/r06def/pmc/summary.txt | sed 's/^TRAFFIC_JSON //' | python3 -c "import sys,json; print(json.dumps(json.loads(sys.stdin.read()), indent=1))" > profiles/r06_pmc_traffic.json || exit 1
cp profiles/r06_pmc_traffic.json $OUT/pmc_traffic.json
grep -v TRAFFIC_JSON gpurun_out/r06def/pmc/summary.txt | grep -v "k_table_probe\|k_table_export" > $OUT/pmc_summary.txt
echo "== bench, driver's command"
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err; rc=$?; cut -c1-200 $OUT/bench_k20.json; bad $rc && exit 1
echo "== bench, default command"
timeout -k 10 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; rc=$?; cut -c1-200 $OUT/bench.json; bad $rc && exit 1
timeout -k 10 300 python3 tools/archive/exp_adapters.py 2>/dev/null | tee $OUT/adapters.json
timeout -k 10 300 python3 tools/archive/exp_unfused.py 2>/dev/null | tee $OUT/four_call.jsonl
exit 0
