#!/usr/bin/env bash
# round 6: a longer fuzz run of the parity check on the final sources (shipped library, then the measurement build)
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r06u; mkdir -p $OUT
export TMPDIR=/tmp
FUZZ_PRODUCT=1 timeout -k 10 500 python3 tests/fuzz_parity.py 6101 40 2>/dev/null | tail -n 41 | cut -c1-250 > $OUT/fuzz_product_40.txt; tail -n 2 $OUT/fuzz_product_40.txt
timeout -k 10 500 python3 tests/fuzz_parity.py 6102 40 2>/dev/null | tail -n 41 | cut -c1-250 > $OUT/fuzz_experiments_40.txt; tail -n 2 $OUT/fuzz_experiments_40.txt
