#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05u; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
rc=$?; tail -n 8 $OUT/pytest_gpu.txt | cut -c1-300; echo "pytest rc $rc"
