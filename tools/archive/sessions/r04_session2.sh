#!/usr/bin/env bash
# Round 4, session 2: what a launch's fixed cost is made of; the two-stream 4-call experiment; the
# growth / high-load / aliasing tests.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04b; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "grows or high_load or alias or row_cache_and or chunked or checkpoint or resume or import" > $OUT/pytest_subset.log 2>&1; rc=$?
tail -n 15 $OUT/pytest_subset.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 tools/archive/exp_intercept.py 2>$OUT/err.log | tee $OUT/intercept.jsonl || { tail -5 $OUT/err.log; exit 1; }
timeout -k 10 600 python3 tools/archive/exp_two_streams.py 2>$OUT/err2.log | tee $OUT/two_streams.jsonl || { tail -5 $OUT/err2.log; exit 1; }
