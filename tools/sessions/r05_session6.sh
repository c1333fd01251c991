#!/usr/bin/env bash
# round 5, session 6: the new tests, the step time against the table's load (pre-filled), 4x4 and 5x5
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05f; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "bound_to_their_table or full_table_stays or row_cache or four_call or high_load" > $OUT/pytest_new.txt 2>&1
rc=$?; tail -n 12 $OUT/pytest_new.txt | cut -c1-300; echo "pytest rc $rc"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python3 tools/exp_load_curve_prefilled.py 30 4 > $OUT/load_curve_prefilled.jsonl 2> $OUT/load_curve.err; echo "rc $?"; cut -c1-260 $OUT/load_curve_prefilled.jsonl
timeout -k 10 600 python3 tools/exp_load_curve_prefilled.py 30 5 > $OUT/load_curve_prefilled_5x5.jsonl 2> $OUT/load_curve5.err; echo "rc $?"; cut -c1-260 $OUT/load_curve_prefilled_5x5.jsonl
