#!/usr/bin/env bash
# The step loop with the new state's claim issued ahead of the probe and the new episode's row probed
# together with s': parity (the whole GPU suite), then old (256 lanes) / new (256) / new (512 lanes) on the
# driver's command, the intercept, and 65 536 boards.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04h; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?
tail -n 12 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
for r in 1 2; do bash tools/archive/sessions/r03_ab_lib.sh "--steps 20 --warmup 5" libq2048_old256.so product libq2048_new512.so 2>&1 | tee -a $OUT/restructure_driver.txt; done
for v in libq2048_old256.so product libq2048_new512.so; do
  if [ "$v" = product ]; then unset Q2048_LIB_PATH; else export Q2048_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/$v; fi
  INTERCEPT_ONLY="learning, row cache" timeout -k 10 300 python3 tools/archive/exp_intercept.py 2>>$OUT/err.log | head -1 | sed "s/^/$v /" | tee -a $OUT/restructure_intercept.txt
  INTERCEPT_BOARDS=65536 INTERCEPT_ONLY="learning, row cache" timeout -k 10 300 python3 tools/archive/exp_intercept.py 2>>$OUT/err.log | head -1 | sed "s/^/$v /" | tee -a $OUT/restructure_intercept.txt
done
unset Q2048_LIB_PATH
