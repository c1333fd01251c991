#!/usr/bin/env bash
# rocprofv3 counter passes on a short bench run (each --pmc set in its own pass).
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG/pmc
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 bench.py --cpu-seconds 0"
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_EA0_ATOMIC_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  echo "== pass $i: $set"
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pass$i" -- $BENCH > "$OUT/pass$i.json" 2> "$OUT/pass$i.err"
  rc=$?; echo "rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done
python3 tools/pmc_summary.py "$OUT" 1048576 64 | tee "$OUT/summary.txt"
