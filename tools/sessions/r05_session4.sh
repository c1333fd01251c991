#!/usr/bin/env bash
# round 5, session 4: growth off the critical path with retired tables and the early commit of a prefetched table;
# train.py on memory that has been free for a while (a pause before every run: memory released by the process before
# is wiped by the driver for seconds, and a mapping made meanwhile waits for it -- profiles/r05_vmm_wipe.txt)
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05d; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "grows or train_default or chunked or placement or checkpoint or resume" > $OUT/pytest_growth.txt 2>&1
rc=$?; tail -n 25 $OUT/pytest_growth.txt | cut -c1-300; echo "pytest rc $rc"; [ $rc -eq 0 ] || exit $rc
T="timeout -k 10 600 python3 train.py"
filt() { grep -v "^epoch [0-9]*[1-9]/" | grep -v amdgpu.ids | tail -n 24; }
sleep 20
echo "== 262144 x 100, default (async growth, 2^30 first, prefetch)"
$T --num-envs 262144 --episodes 100 --log $OUT/train_262144x100_growing.csv 2>&1 | filt | tee $OUT/train_262144x100_growing.log
sleep 20
echo "== 1048576 x 20, default"
$T --num-envs 1048576 --episodes 20 --log $OUT/train_1048576x20_growing.csv 2>&1 | filt | tee $OUT/train_1048576x20_growing.log
sleep 20
echo "== 262144 x 100, first capacity 2^28"
$T --num-envs 262144 --episodes 100 --initial-capacity-log2 28 --log $OUT/train_262144x100_growing_from28.csv 2>&1 | filt | tee $OUT/train_262144x100_growing_from28.log
sleep 20
echo "== 262144 x 100, fixed 2^32"
$T --num-envs 262144 --episodes 100 --capacity-log2 32 --log $OUT/train_262144x100_fixed32.csv 2>&1 | filt | tee $OUT/train_262144x100_fixed32.log
sleep 20
echo "== 262144 x 100, sync growth from 2^28 (round 4's way)"
$T --num-envs 262144 --episodes 100 --initial-capacity-log2 28 --growth sync --log $OUT/train_262144x100_sync_from28.csv 2>&1 | filt | tee $OUT/train_262144x100_sync_from28.log
echo "== 262144 x 100, default, right after the run before (no pause: freshly released memory)"
$T --num-envs 262144 --episodes 100 --log $OUT/train_262144x100_growing_nopause.csv 2>&1 | filt | tee $OUT/train_262144x100_growing_nopause.log
