#!/usr/bin/env bash
# 512-lane workgroups against 256 (product): ten alternations of the driver's command, three of the default one.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04g; mkdir -p $OUT
for r in 1 2 3; do bash tools/archive/sessions/r03_ab_lib.sh "--steps 20 --warmup 5" product libq2048_b512.so 2>&1 | tee -a $OUT/b512_driver.txt; done
bash tools/archive/sessions/r03_ab_lib.sh "--steps 256 --warmup 64" product libq2048_b512.so 2>&1 | sed "s/^/default /" | tee -a $OUT/b512_default.txt
bash tools/archive/sessions/r03_ab_lib.sh "--steps 20 --warmup 5 --board-size 5" product libq2048_b512.so 2>&1 | sed "s/^/5x5 /" | tee -a $OUT/b512_5x5.txt
