#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05t; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "two_ranks or c_host or growth_in_flight or grows" > $OUT/pytest.txt 2>&1
rc=$?; tail -n 12 $OUT/pytest.txt | cut -c1-300; echo "pytest rc $rc"
