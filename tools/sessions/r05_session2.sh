#!/usr/bin/env bash
# round 5, session 2: chunks of at most 32 MiB (created in microseconds; 64 MiB and larger take milliseconds each,
# session 1), by count; and the mapping in a host thread next to a stream of launches
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05b; mkdir -p $OUT
export TMPDIR=/tmp
V=tools/variants/exp_vmm_cost
{
  timeout -k 10 120 $V map 128 32 &&
  timeout -k 10 120 $V map 128 16 &&
  timeout -k 10 120 $V map 128 8 &&
  timeout -k 10 120 $V map 128 32 32 &&
  timeout -k 10 120 $V map 64 32 16 &&
  timeout -k 10 120 $V map 64 16 &&
  timeout -k 10 120 $V map 32 8 &&
  timeout -k 10 120 $V bg 32 32 8 &&
  timeout -k 10 120 $V bg 128 32 32 &&
  timeout -k 10 120 $V bg 128 16 32
} > $OUT/vmm_cost2.txt 2>&1
echo "vmm_cost rc $?"; grep -v amdgpu.ids $OUT/vmm_cost2.txt | cut -c1-250
