#!/usr/bin/env bash
# Multi-rank rehearsal on the one leased GPU (gloo): bench.py --gpus 2 and --gpus 4 start their own ranks; one
# statistics collective per region (all-gather), one MAX collective after the last region.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04x; mkdir -p $OUT
export Q2048_DIST_BACKEND=gloo
timeout -k 10 500 python3 bench.py --gpus 2 --steps 20 --warmup 5 --boards-per-gpu 524288 --cpu-seconds 2 > $OUT/bench_2rank.json 2> $OUT/bench_2rank.err; echo "rc=$?"; cut -c1-300 $OUT/bench_2rank.json
timeout -k 10 500 python3 bench.py --gpus 4 --steps 20 --warmup 5 --boards-per-gpu 262144 --cpu-seconds 2 > $OUT/bench_4rank.json 2> $OUT/bench_4rank.err; echo "rc=$?"; cut -c1-300 $OUT/bench_4rank.json
unset Q2048_DIST_BACKEND
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --no-companions > $OUT/bench_1rank_rccl.json 2> $OUT/bench_1rank_rccl.err; echo "rc=$?"; cut -c1-300 $OUT/bench_1rank_rccl.json
