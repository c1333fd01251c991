#!/usr/bin/env bash
# Round 3, final evidence on the final sources: session 8a (suite, smoke, benches, kernel traces), then the
# PMC passes of the driver's command, of the default command at its own table size, and of 5x5.
set -u
TAG=${1:-r03m}
cd "$GRAFT_REPO_ROOT"
bash tools/archive/sessions/r03_session8a.sh $TAG || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
for spec in "k20:--steps 20 --warmup 5" "def:--cap-log2 32" "5x5:--board-size 5 --cap-log2 32"; do
  name=${spec%%:*}; args=${spec#*:}
  bash tools/pmc_session.sh ${TAG}_$name $args > "$OUT/pmc_$name.log" 2>&1; echo "pmc $name rc=$?"
  grep TRAFFIC_JSON "$OUT/pmc_$name.log" | cut -c1-200
  find "gpurun_out/${TAG}_$name/pmc" -name "*.csv" -size +1M -delete
done
