#!/usr/bin/env bash
# Soak: train.py on growing tables with the row count verified at every report -- 5x5 boards, the
# deterministic step, strict TD writes -- and evaluate.py on a saved growing-table run.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04t; mkdir -p $OUT
export TMPDIR=/tmp
echo "== 5x5, 262 144 envs x 12 episodes"
timeout -k 10 500 python3 train.py --board-size 5 --num-envs 262144 --episodes 12 --log $OUT/train5.csv 2>&1 | grep -v "^epoch [0-9]*[1-9]/" | tail -n 12 | tee $OUT/train_5x5_growing.log
echo "== deterministic step, 65 536 envs x 20 episodes"
timeout -k 10 500 python3 train.py --deterministic --num-envs 65536 --episodes 20 --log $OUT/traind.csv 2>&1 | grep -v "^epoch [0-9]*[1-9]/" | tail -n 8 | tee $OUT/train_deterministic_growing.log
echo "== strict TD, 262 144 envs x 14 episodes, saved (> 4 GiB of rows), evaluated"
timeout -k 10 500 python3 train.py --strict-td --num-envs 262144 --episodes 14 --save $OUT/q.pt --log $OUT/trains.csv 2>&1 | grep -v "^epoch [0-9]*[1-9]/" | tail -n 8 | tee $OUT/train_strict_growing.log
timeout -k 10 300 python3 evaluate.py --model $OUT/q.pt --num-envs 65536 --episodes 2 2>&1 | tail -n 3 | tee $OUT/evaluate.log
rm -f $OUT/q.pt $OUT/q.pt.rank*
