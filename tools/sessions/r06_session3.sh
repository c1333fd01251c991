#!/usr/bin/env bash
# round 6, session 3: growth-path hardening tests, the C host example, train.py on a table that freezes, bench with the frozen companion
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r06c; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -s -k "commit_contract or grows_like or c_host_program or process_exit or finds_no_room or closed_key or full_table or freezes" > $OUT/pytest_new.txt 2>&1
rc=$?; grep -a "^\[" $OUT/pytest_new.txt | cut -c1-300; tail -n 6 $OUT/pytest_new.txt | cut -c1-300; echo "pytest(new) rc $rc"; [ $rc -eq 0 ] || exit $rc
# (a) of the review: a 5x5 run that outgrows the largest table
timeout -k 10 900 python3 train.py --num-envs 262144 --episodes 60 --board-size 5 --log $OUT/train_5x5.csv > $OUT/train_5x5_freeze.log 2>&1
rc=$?; tail -n 12 $OUT/train_5x5_freeze.log | cut -c1-400; echo "train rc $rc"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err
rc=$?; cut -c1-1500 $OUT/bench_k20.json; tail -n 5 $OUT/bench_k20.err; echo "bench rc $rc"
