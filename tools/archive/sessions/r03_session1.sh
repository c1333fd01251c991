#!/usr/bin/env bash
# Round 3, session 1 (no code changes yet): per-kernel numbers of the batched 4-call API, and the
# table-placement effect seen through per-instance (XCD x L2 channel) write-request counters.
set -u
TAG=${1:-r03a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
echo "== 4-call API under rocprofv3 --kernel-trace --stats"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/four_call" -- \
    python3 tools/archive/exp_unfused.py > "$OUT/four_call.jsonl" 2> "$OUT/four_call.err"; rc=$?
echo "rc=$rc"; cat "$OUT/four_call.jsonl"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
find "$OUT/four_call" -name "*kernel_stats.csv" | while read -r f; do cp "$f" "$OUT/four_call_kernel_stats.csv"; head -n 14 "$f"; done
# keep the trace small: only the columns the per-grid summary needs
find "$OUT/four_call" -name "*kernel_trace.csv" | while read -r f; do
  python3 - "$f" "$OUT/four_call_by_grid.txt" <<'PY'
import csv, sys, collections, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-48:]
    acc[(name, int(r["Grid_Size"]) if "Grid_Size" in r else int(r.get("Grid_Size_X", 0)))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(sys.argv[2], "w") as f:
    for (n, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        line = f"{n:50s} grid {g:10d} n={len(v):5d} median_us={statistics.median(v):9.2f} mean_us={sum(v)/len(v):9.2f} total_ms={sum(v)/1e3:9.2f}"
        print(line); f.write(line + "\n")
PY
done
rm -rf "$OUT/four_call"

echo "== placement: per-instance write-request counters (json)"
BIN=$GRAFT_REPO_ROOT/tools/variants/exp_place
OPS="a0:32 s0 a1:32 s1 a2:32 s2 a3:32 s3 a4:32 s4 a5:32 s5 a6:32 s6"
i=0
for set in "TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ" "TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_LEVEL" "TCC_REQ TCC_TAG_STALL"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format json -d "$OUT/place$i" -- $BIN $OPS > "$OUT/place$i.txt" 2> "$OUT/place$i.err"
  rc=$?; echo "place pass $i rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
  grep us/step "$OUT/place$i.txt" | awk '{printf "%s ", $2}'; echo
  find "$OUT/place$i" -name "*.json" | while read -r f; do ls -la "$f"; gzip -9 "$f"; done
done
du -sh "$OUT"
