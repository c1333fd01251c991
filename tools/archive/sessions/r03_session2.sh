#!/usr/bin/env bash
# Round 3, session 2: the whole GPU suite on the reworked deterministic step (incl. 1 Mi boards vs
# the oracle), then the deterministic mode's timing.
set -u
TAG=${1:-r03b}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
echo "== det tests first"
timeout -k 10 1000 python -m pytest tests -m gpu -q -x -k "deterministic or flag_bits" -rA > "$OUT/pytest_det.log" 2>&1; rc=$?
tail -n 30 "$OUT/pytest_det.log"; echo "rc=$rc"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
echo "== the rest"
timeout -k 10 1000 python -m pytest tests -m gpu -q -k "not deterministic and not flag_bits" > "$OUT/pytest_rest.log" 2>&1; rc=$?
tail -n 15 "$OUT/pytest_rest.log"; echo "rc=$rc"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
echo "== det timing"
timeout -k 10 300 python tools/archive/exp_det.py > "$OUT/det.jsonl" 2> "$OUT/det.err"; echo "rc=$?"; cat "$OUT/det.jsonl"
