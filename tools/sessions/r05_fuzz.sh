#!/usr/bin/env bash
# round 5: the confidence run of the parity check on the SHIPPED library (final sources), two seeds x 12 cases
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05y; mkdir -p $OUT
export TMPDIR=/tmp
for s in 51 52; do FUZZ_PRODUCT=1 timeout -k 10 500 python3 tests/fuzz_parity.py $s 12 2>&1 | grep -v amdgpu.ids | tee -a $OUT/fuzz_product_seeds51_52.txt | tail -n 2; done
