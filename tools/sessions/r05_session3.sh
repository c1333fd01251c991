#!/usr/bin/env bash
# round 5, session 3: growth off the critical path -- parity tests, then train.py on a growing table
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05c; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "grows or train_default or chunked or placement or high_load or checkpoint or resume" > $OUT/pytest_growth.txt 2>&1
rc=$?; tail -n 25 $OUT/pytest_growth.txt | cut -c1-300; echo "pytest rc $rc"; [ $rc -eq 0 ] || exit $rc
T="timeout -k 10 600 python3 train.py"
filt() { grep -v "^epoch [0-9]*[1-9]/" | grep -v amdgpu.ids | tail -n 24; }
echo "== 262144 x 100, default (async growth, 2^30 first, prefetch)"
$T --num-envs 262144 --episodes 100 --log $OUT/train_262144x100_growing.csv 2>&1 | filt | tee $OUT/train_262144x100_growing.log
echo "== 262144 x 100, first capacity 2^28"
$T --num-envs 262144 --episodes 100 --initial-capacity-log2 28 --log $OUT/train_262144x100_growing_from28.csv 2>&1 | filt | tee $OUT/train_262144x100_growing_from28.log
echo "== 262144 x 100, sync growth from 2^28"
$T --num-envs 262144 --episodes 100 --initial-capacity-log2 28 --growth sync --log $OUT/train_262144x100_sync_from28.csv 2>&1 | filt | tee $OUT/train_262144x100_sync_from28.log
echo "== 262144 x 100, fixed 2^32"
$T --num-envs 262144 --episodes 100 --capacity-log2 32 --log $OUT/train_262144x100_fixed32.csv 2>&1 | filt | tee $OUT/train_262144x100_fixed32.log
echo "== 1048576 x 20, default"
$T --num-envs 1048576 --episodes 20 --log $OUT/train_1048576x20_growing.csv 2>&1 | filt | tee $OUT/train_1048576x20_growing.log
echo "== freshly released memory: map 128 GiB right after a process that held 128 GiB, and 12 s later"
V=tools/variants/exp_vmm_cost
{ timeout -k 10 120 $V map 128 32 32 && timeout -k 10 120 $V map 128 32 32 && sleep 12 && timeout -k 10 120 $V map 128 32 32 && sleep 12 && timeout -k 10 120 $V bg 128 32 32; } > $OUT/vmm_wipe.txt 2>&1
grep -v amdgpu.ids $OUT/vmm_wipe.txt | cut -c1-230
