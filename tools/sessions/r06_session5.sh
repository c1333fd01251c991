#!/usr/bin/env bash
# round 6, session 5: the tests added after the closing sessions
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r06e; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -s -k "really_fails or commit_contract or legitimate or full_table or finds_no_room or falls_back" > $OUT/pytest_new.txt 2>&1
rc=$?; grep -a "^\[" $OUT/pytest_new.txt | cut -c1-300; tail -n 5 $OUT/pytest_new.txt | cut -c1-300; echo "pytest(new) rc $rc"
