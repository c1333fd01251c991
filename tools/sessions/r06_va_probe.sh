#!/usr/bin/env bash
# round 6: where reservations land after a large hipFree, and whether an address hint in a private region is honoured
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r06v; mkdir -p $OUT
timeout -k 10 120 tools/variants/va_hint_probe 64 | tee $OUT/va_hint_probe.jsonl
timeout -k 10 120 tools/variants/va_hint_probe 200 | tee -a $OUT/va_hint_probe.jsonl
