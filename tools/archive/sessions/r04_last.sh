#!/usr/bin/env bash
# Round 4, closing session on the final tree: the whole GPU suite, smoke(), the timeline with the shipped
# workgroup size, three fuzz seeds (measurement build and product library).
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04y; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?
tail -n 5 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2 | tee $OUT/smoke.log
timeout -k 10 200 python3 tools/archive/exp_timeline.py > $OUT/timeline_512.jsonl 2>$OUT/err.log; cut -c1-400 $OUT/timeline_512.jsonl | head -3; tail -2 $OUT/err.log
timeout -k 10 400 python3 tests/fuzz_parity.py 12 10 2>>$OUT/err.log | tail -n 1 | tee $OUT/fuzz_seed12.txt
FUZZ_PRODUCT=1 timeout -k 10 400 python3 tests/fuzz_parity.py 13 10 2>>$OUT/err.log | tail -n 1 | tee $OUT/fuzz_product_seed13.txt
