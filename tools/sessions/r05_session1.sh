#!/usr/bin/env bash
# round 5, session 1: what the virtual-memory calls behind a growth cost (by size / chunk / what is held / in a
# host thread next to kernel launches), and the library-free reproducer of the re-used address range
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05a; mkdir -p $OUT
export TMPDIR=/tmp
V=tools/variants/exp_vmm_cost
{
  timeout -k 10 120 $V map 32 32 &&
  timeout -k 10 120 $V map 64 64 &&
  timeout -k 10 120 $V map 128 64 &&
  timeout -k 10 120 $V map 128 256 &&
  timeout -k 10 120 $V map 128 1024 &&
  timeout -k 10 120 $V map 128 64 64 &&
  timeout -k 10 120 $V map 128 1024 64 &&
  timeout -k 10 120 $V malloc 128 &&
  timeout -k 10 120 $V malloc 32 &&
  timeout -k 10 120 $V bg 32 32 8 &&
  timeout -k 10 120 $V bg 128 64 32 &&
  timeout -k 10 120 $V bg 128 1024 32
} > $OUT/vmm_cost.txt 2>&1
echo "vmm_cost rc $?"; grep -v amdgpu.ids $OUT/vmm_cost.txt | cut -c1-250
R=tools/variants/va_reuse_repro
{
  timeout -k 10 200 $R 0 27 8 4 &&
  timeout -k 10 200 $R 1 27 8 4 &&
  timeout -k 10 200 $R 2 27 8 4 &&
  timeout -k 10 200 $R 3 27 8 4
} > $OUT/va_reuse_repro.txt 2>&1
echo "va_reuse rc $?"; grep -v amdgpu.ids $OUT/va_reuse_repro.txt | cut -c1-220
