#!/usr/bin/env bash
# Soak 3: the other train.py / evaluate.py paths on growing tables.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04u; mkdir -p $OUT
export TMPDIR=/tmp
run() { echo "== $*"; timeout -k 10 400 "$@" 2>&1 | grep -v "amdgpu.ids\|^epoch [0-9]*[1-9]/" | tail -n 6 | cut -c1-400; echo "rc=${PIPESTATUS[0]}"; }
run python3 train.py --board-size 5 --num-envs 65536 --episodes 6 --save /tmp/q5.pt --log $OUT/t5.csv
run python3 evaluate.py --model /tmp/q5.pt --num-envs 16384 --episodes 2
run python3 train.py --num-envs 65536 --episodes 12 --stop-epoch 6 --save /tmp/qa.pt --log $OUT/ta.csv
run python3 train.py --num-envs 65536 --episodes 12 --resume /tmp/qa.pt --save /tmp/qb.pt --log $OUT/tb.csv
run python3 train.py --env-profile nopenalty --num-envs 65536 --episodes 6 --log $OUT/tn.csv
run python3 train.py --agent row-tuple --num-envs 65536 --episodes 10 --log $OUT/tr.csv
Q2048_DIST_BACKEND=gloo run python3 train.py --gpus 2 --num-envs 32768 --episodes 8 --log $OUT/t2.csv
rm -f /tmp/q5.pt /tmp/qa.pt /tmp/qb.pt /tmp/qa.pt.rank* /tmp/qb.pt.rank*
