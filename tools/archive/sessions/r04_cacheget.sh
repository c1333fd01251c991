#!/usr/bin/env bash
# cache_get with both halves of a record requested before the key compare (one round trip instead of two):
# launch fits, the 4-call loop, the driver's command; product against the variant build.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04cg; mkdir -p $OUT
for v in product libq2048_cacheget.so product libq2048_cacheget.so; do
  if [ "$v" = product ]; then unset Q2048_LIB_PATH; else export Q2048_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/$v; fi
  INTERCEPT_ONLY="learning, row cache" timeout -k 10 300 python3 tools/archive/exp_intercept.py 2>>$OUT/err.log | head -1 | sed "s/^/$v /" | tee -a $OUT/cacheget_intercept.txt
  timeout -k 10 300 python3 tools/archive/exp_unfused.py 2>>$OUT/err.log | grep "4-call" | sed "s/^/$v /" | tee -a $OUT/cacheget_four_call.txt
done
unset Q2048_LIB_PATH
for r in 1 2; do bash tools/archive/sessions/r03_ab_lib.sh "--steps 20 --warmup 5" product libq2048_cacheget.so 2>&1 | tee -a $OUT/cacheget_driver.txt; done
