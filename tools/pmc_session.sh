#!/usr/bin/env bash
# rocprofv3 counter passes on a short bench run (each --pmc set in its own pass, never combined
# with a trace domain other than --kernel-trace).  Usage: bash tools/pmc_session.sh <tag> [bench args]
set -u
TAG=${1:-r02}
shift || true
OUT=gpurun_out/$TAG/pmc
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 bench.py --cpu-seconds 0 --no-companions --repeats 2 $*"
echo "$BENCH" > "$OUT/command.txt"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_EA0_ATOMIC_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RD_UNCACHED_32B_sum" \
           "TCC_EA0_WRREQ_64B_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_WRREQ_ATOMIC_DRAM_32B_sum" \
           "TCC_WRITEBACK_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum" \
           "TCC_READ_sum TCC_WRITE_sum TCC_ATOMIC_sum TCC_REQ_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU"; do
  i=$((i+1))
  echo "== pass $i: $set"
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pass$i" -- $BENCH > "$OUT/pass$i.json" 2> "$OUT/pass$i.err"
  rc=$?; echo "rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done
python3 tools/pmc_summary.py "$OUT" "$OUT/pass1.json" | tee "$OUT/summary.txt"
