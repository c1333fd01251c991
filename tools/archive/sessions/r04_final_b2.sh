#!/usr/bin/env bash
# The two bench lines again (the 5x5 companion now on the main line's table size).
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04z; mkdir -p $OUT
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err; rc=$?; cut -c1-200 $OUT/bench_k20.json; [ $rc -eq 0 ] || exit 1
timeout -k 10 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; rc=$?; cut -c1-200 $OUT/bench.json; [ $rc -eq 0 ] || exit 1
