#!/usr/bin/env bash
# round 6: what the driver runs at round end, in its order -- smoke(), then the bench command, timed
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r06r; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
SECONDS=0
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err; echo "bench rc $? wall ${SECONDS} s"
cut -c1-160 $OUT/bench_k20.json
