#!/usr/bin/env bash
# round 6: the run VERDICT r5 (weak #2) named -- `python train.py --num-envs 1048576 --episodes 1000` (BASELINE configs[2]'s
# batch at configs[0]'s episode count, 1.7e11 env-steps): it outgrows the largest table after 2 % of the run, closes the key
# set at load 0.5 and goes on at the young table's speed.  Every 50th epoch line is kept.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r06r; mkdir -p $OUT
export TMPDIR=/tmp
S=$SECONDS
timeout -k 10 1000 python3 train.py --num-envs 1048576 --episodes 1000 --log $OUT/train_1048576x1000.csv 2>&1 \
  | grep -v amdgpu.ids | grep -v "^epoch [0-9]*[1-9]/" | grep -v "^epoch [0-9]*[1234678]0/" | cut -c1-260 | tee $OUT/train_1048576x1000.log
echo "wall seconds: $((SECONDS - S))" | tee -a $OUT/train_1048576x1000.log
head -c 4000 $OUT/train_1048576x1000.csv > $OUT/train_1048576x1000_head.csv; tail -n 5 $OUT/train_1048576x1000.csv >> $OUT/train_1048576x1000_head.csv
rm -f $OUT/train_1048576x1000.csv
exit 0
