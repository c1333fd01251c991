#!/usr/bin/env bash
# PMC passes on the default command at the default command's own table size (2^32 slots: with
# --repeats 2 the sizing rule would otherwise pick 2^31)
set -u
TAG=${1:-r03k}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
bash tools/pmc_session.sh ${TAG}_def --cap-log2 32 > "$OUT/pmc_def.log" 2>&1; echo "rc=$?"
grep TRAFFIC_JSON "$OUT/pmc_def.log" | cut -c1-300
find "gpurun_out/${TAG}_def/pmc" -name "*.csv" -size +1M -delete
bash tools/pmc_session.sh ${TAG}_5x5 --board-size 5 --cap-log2 32 > "$OUT/pmc_5x5.log" 2>&1; echo "rc=$?"
find "gpurun_out/${TAG}_5x5/pmc" -name "*.csv" -size +1M -delete
