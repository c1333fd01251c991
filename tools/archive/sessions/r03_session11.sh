#!/usr/bin/env bash
# Round 3, evidence after the 5x5 core rewrite: session 10 (suite, smoke, benches, traces, PMC passes,
# 4-call trace, deterministic step) + the single-step kernels' timing + a fuzz run of the parity check.
set -u
TAG=${1:-r03r}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
bash tools/archive/sessions/r03_session10.sh $TAG || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
echo "== single-step kernels"
timeout -k 10 200 python3 tools/archive/exp_env_step.py > "$OUT/env_step.jsonl" 2> /dev/null; grep '"boards_per_thread": "default"' "$OUT/env_step.jsonl"
echo "== fuzz parity, seed 5"
timeout -k 10 500 python3 tests/fuzz_parity.py 5 16 > "$OUT/fuzz.log" 2>&1; echo "rc=$?"; tail -n 3 "$OUT/fuzz.log" | cut -c1-300
