#!/usr/bin/env bash
# 768- and 1024-lane workgroups against the shipped 512 (big batches): alternating pairs of the driver's command.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04s; mkdir -p $OUT
for r in 1 2; do bash tools/archive/sessions/r03_ab_lib.sh "--steps 20 --warmup 5" product libq2048_B768.so libq2048_B1024.so 2>&1 | tee -a $OUT/block768_driver.txt; done
bash tools/archive/sessions/r03_ab_lib.sh "--steps 256 --warmup 64" product libq2048_B768.so 2>&1 | sed "s/^/default /" | tee -a $OUT/block768_default.txt
