cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04t; mkdir -p $OUT
export TMPDIR=/tmp
echo "== strict TD, 262 144 envs x 14 episodes, saved (> 4 GiB of rows), evaluated"
timeout -k 10 700 python3 train.py --strict-td --num-envs 262144 --episodes 14 --save /tmp/q2048_soak.pt --log $OUT/trains.csv 2>&1 | grep -v "^epoch [0-9]*[1-9]/" | tail -n 8 | tee $OUT/train_strict_growing.log
ls -la /tmp/q2048_soak.pt | tee -a $OUT/train_strict_growing.log
timeout -k 10 400 python3 evaluate.py --model /tmp/q2048_soak.pt --num-envs 65536 --episodes 2 2>&1 | tail -n 3 | tee $OUT/evaluate.log
rm -f /tmp/q2048_soak.pt
