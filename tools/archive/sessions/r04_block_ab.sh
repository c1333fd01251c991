#!/usr/bin/env bash
# Workgroup size of the fused rollout (64 / 128 / 256 = product / 512 lanes): a launch's fixed cost is its
# drain -- blocks of one launch take 300-400 us for the same 20 steps, and a block's slots are free only
# when its slowest wave is done (profiles/r04_launch_timeline.jsonl).  launch(S) fits, then the driver's command.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04f; mkdir -p $OUT
export TMPDIR=/tmp
for v in product libq2048_b64.so libq2048_b128.so libq2048_b512.so product libq2048_b64.so; do
  if [ "$v" = product ]; then unset Q2048_LIB_PATH; else export Q2048_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/$v; fi
  INTERCEPT_ONLY="learning, row cache" timeout -k 10 300 python3 tools/archive/exp_intercept.py 2>>$OUT/err.log | head -1 | sed "s/^/$v /" | tee -a $OUT/block_intercept.txt
done
unset Q2048_LIB_PATH
bash tools/archive/sessions/r03_ab_lib.sh "--steps 20 --warmup 5" product libq2048_b64.so libq2048_b128.so libq2048_b512.so 2>&1 | tee -a $OUT/block_driver_ab.txt
bash tools/archive/sessions/r03_ab_lib.sh "--steps 20 --warmup 5 --board-size 5" product libq2048_b64.so libq2048_b128.so 2>&1 | sed "s/^/5x5 /" | tee -a $OUT/block_driver_ab.txt
