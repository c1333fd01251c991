#!/usr/bin/env bash
# round 6, session 6: tables in a private region of the address space -- the growth tests first (incl. the regression test of
# the round's GPU fault), then the whole suite, then the driver's bench command (placement must not have cost anything)
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r06g; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -s -k "really_fails or commit_contract or grows_like or chunked or placements or process_exit or spanning" > $OUT/pytest_new.txt 2>&1
rc=$?; grep -a "^\[" $OUT/pytest_new.txt | cut -c1-300; tail -n 5 $OUT/pytest_new.txt | cut -c1-300; echo "pytest(new) rc $rc"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
rc=$?; tail -n 4 $OUT/pytest_gpu.txt | cut -c1-300; echo "pytest(all) rc $rc"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err; echo "bench rc $?"; cut -c1-200 $OUT/bench_k20.json
