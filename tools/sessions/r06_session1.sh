#!/usr/bin/env bash
# round 6, session 1: the closed-key-set tests, the whole GPU suite, the load curve with the key set open / closed
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r06a; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -s -k "closed_key or full_table or freezes or flag_bits or table_full" > $OUT/pytest_new.txt 2>&1
rc=$?; grep -a "^\[" $OUT/pytest_new.txt | cut -c1-300; tail -n 6 $OUT/pytest_new.txt | cut -c1-300; echo "pytest(new) rc $rc"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
rc=$?; tail -n 8 $OUT/pytest_gpu.txt | cut -c1-300; echo "pytest(all) rc $rc"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python3 tools/exp_load_curve_prefilled.py 30 4 > $OUT/load_curve_frozen.jsonl 2> $OUT/load_curve.err; echo "rc $?"; cut -c1-420 $OUT/load_curve_frozen.jsonl
timeout -k 10 600 python3 tools/exp_load_curve_prefilled.py 30 5 > $OUT/load_curve_frozen_5x5.jsonl 2> $OUT/load_curve5.err; echo "rc $?"; cut -c1-420 $OUT/load_curve_frozen_5x5.jsonl
