#!/usr/bin/env bash
# round 6, session 4: visit rows -- the GPU suite, the 5x5 run that outgrows the largest table, the learning A/B, the wipe A/B
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r06d; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
rc=$?; tail -n 6 $OUT/pytest_gpu.txt | cut -c1-300; echo "pytest(all) rc $rc"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python3 train.py --num-envs 262144 --episodes 60 --board-size 5 --log $OUT/train_5x5.csv > $OUT/train_5x5_freeze.log 2>&1
rc=$?; grep -a "frozen\|grew" $OUT/train_5x5_freeze.log | cut -c1-300; tail -n 4 $OUT/train_5x5_freeze.log | cut -c1-400; echo "train rc $rc"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python3 tools/exp_learning_ab.py --modes store/64,cas,det,frozen --seeds 3 > $OUT/learning_ab.jsonl 2> $OUT/learning_ab.err
rc=$?; cut -c1-420 $OUT/learning_ab.jsonl; tail -n 3 $OUT/learning_ab.err; echo "learning rc $rc"
timeout -k 10 600 python3 tools/exp_wipe_ab.py > $OUT/wipe_ab.jsonl 2> $OUT/wipe_ab.err
rc=$?; cut -c1-300 $OUT/wipe_ab.jsonl; tail -n 3 $OUT/wipe_ab.err; echo "wipe rc $rc"
