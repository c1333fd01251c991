#!/usr/bin/env bash
# The episode boundary of the fused step loop restructured (an ending lane finishes its episode before the
# step's one probe and looks its NEW state up there; a state without a row is claimed at the top of the step):
# the whole GPU suite, then previous build / this one on the driver's command, the default one, 5x5, the launch
# fit at 1 Mi and at 65 536 boards, and the timeline.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04r; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?
tail -n 6 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
for r in 1 2; do bash tools/archive/sessions/r03_ab_lib.sh "--steps 20 --warmup 5" libq2048_prev.so product 2>&1 | tee -a $OUT/boundary_driver.txt; done
bash tools/archive/sessions/r03_ab_lib.sh "--steps 256 --warmup 64" libq2048_prev.so product 2>&1 | sed "s/^/default /" | tee -a $OUT/boundary_default.txt
bash tools/archive/sessions/r03_ab_lib.sh "--steps 20 --warmup 5 --board-size 5" libq2048_prev.so product 2>&1 | sed "s/^/5x5 /" | tee -a $OUT/boundary_5x5.txt
for v in libq2048_prev.so product; do
  if [ "$v" = product ]; then unset Q2048_LIB_PATH; else export Q2048_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/$v; fi
  INTERCEPT_ONLY="learning, row cache" timeout -k 10 300 python3 tools/archive/exp_intercept.py 2>>$OUT/err.log | head -1 | sed "s/^/$v /" | tee -a $OUT/boundary_intercept.txt
  INTERCEPT_BOARDS=65536 INTERCEPT_ONLY="learning, row cache" timeout -k 10 300 python3 tools/archive/exp_intercept.py 2>>$OUT/err.log | head -1 | sed "s/^/$v /" | tee -a $OUT/boundary_intercept.txt
done
unset Q2048_LIB_PATH
timeout -k 10 300 python3 tools/archive/exp_timeline.py 2>>$OUT/err.log | head -2 | cut -c1-1500 | tee $OUT/timeline.jsonl
