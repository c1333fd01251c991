#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04fa; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?
tail -n 4 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1 | tee $OUT/smoke.log
bash tools/pmc_session.sh r04k20 --steps 20 --warmup 5 2>&1 | grep "TRAFFIC_JSON\|rc=" | cut -c1-200
bash tools/pmc_session.sh r04def --cap-log2 32 2>&1 | grep "TRAFFIC_JSON\|rc=" | cut -c1-200
