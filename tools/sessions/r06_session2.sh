#!/usr/bin/env bash
# round 6, session 2: the gate of table line layout v2 (tools/exp_layout_v2.hip)
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r06b; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 tools/variants/exp_layout_v2 30 20 16 200 1 > $OUT/layout_v2.jsonl 2> $OUT/layout_v2.err; echo "rc $?"
cut -c1-400 $OUT/layout_v2.jsonl; tail -n 3 $OUT/layout_v2.err
