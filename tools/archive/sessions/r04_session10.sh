#!/usr/bin/env bash
# Round 4, session 10: the library with 512-lane workgroups for big batches -- full GPU suite, A/B against the
# 256-lane build, train.py on a growing table, the re-used-address-range debug runs.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04i; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?
tail -n 8 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
for r in 1 2; do bash tools/archive/sessions/r03_ab_lib.sh "--steps 20 --warmup 5" libq2048_old256.so product 2>&1 | tee -a $OUT/block_runtime_choice_ab.txt; done
echo "== train.py on a growing table (no --capacity-log2): 262 144 envs x 100 episodes"
timeout -k 10 900 python3 train.py --num-envs 262144 --episodes 100 --log $OUT/train_262144x100.csv 2>&1 | grep -v "^epoch [0-9]*[1-9]/" | tail -n 40 | tee $OUT/train_262144x100.log
echo "== re-used address ranges (measurement build, round 3's free path)"
for mode in 0 1 2 3; do
  echo "-- Q2048_DEBUG_VA_FREE=$mode" | tee -a $OUT/chunk_debug.txt
  if [ $mode = 0 ]; then timeout -k 10 300 python3 tools/archive/chunk_debug.py 2>&1 | tail -n 8 | tee -a $OUT/chunk_debug.txt
  else Q2048_DEBUG_VA_FREE=$mode timeout -k 10 300 python3 tools/archive/chunk_debug.py 2>&1 | tail -n 30 | cut -c1-400 | tee -a $OUT/chunk_debug.txt; fi
done
