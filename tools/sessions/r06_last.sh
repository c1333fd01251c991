#!/usr/bin/env bash
# round 6, last session: the GPU suite and smoke() on the committed tree
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r06l; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
rc=$?; tail -n 4 $OUT/pytest_gpu.txt | cut -c1-300; echo "pytest rc $rc"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
