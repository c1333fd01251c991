#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04q; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python3 tools/archive/exp_grow.py > $OUT/grow.txt 2>&1; tail -n 16 $OUT/grow.txt | cut -c1-260
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "grows or chunked or placement or high_load" 2>&1 | tail -n 5
echo "== train.py on a growing table: 262 144 envs x 100 episodes"
timeout -k 10 600 python3 train.py --num-envs 262144 --episodes 100 --log $OUT/train_262144x100.csv 2>&1 | grep -v "^epoch [0-9]*[1-9]/" | tail -n 20 | tee $OUT/train_262144x100.log
