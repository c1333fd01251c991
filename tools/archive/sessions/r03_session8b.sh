#!/usr/bin/env bash
# Round 3 evidence, part B: PMC passes (one counter set per pass) on the default command and on the
# driver's command.
set -u
TAG=${1:-r03i}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
bash tools/pmc_session.sh ${TAG}_k20 --steps 20 --warmup 5 > "$OUT/pmc_k20.log" 2>&1; echo "rc=$?"
grep TRAFFIC_JSON "$OUT/pmc_k20.log" | cut -c1-300
find "gpurun_out/${TAG}_k20/pmc" -name "*.csv" -size +1M -delete
bash tools/pmc_session.sh ${TAG}_def > "$OUT/pmc_def.log" 2>&1; echo "rc=$?"
grep TRAFFIC_JSON "$OUT/pmc_def.log" | cut -c1-300
find "gpurun_out/${TAG}_def/pmc" -name "*.csv" -size +1M -delete
