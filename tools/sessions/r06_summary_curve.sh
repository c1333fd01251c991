#!/usr/bin/env bash
# round 6: the step against the table's load with line summaries (the frozen columns), and the long run again
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r06t; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python3 tools/exp_load_curve_prefilled.py 30 4 > $OUT/load_curve_frozen_summaries.jsonl 2> $OUT/load_curve.err; echo "rc $?"
python3 - <<'PY'
import json
for l in open("gpurun_out/r06t/load_curve_frozen_summaries.jsonl"):
    d = json.loads(l); print(round(d["load_before"], 2), d["median_us_per_step"], d["frozen_median_us_per_step"])
PY
S=$SECONDS
timeout -k 10 900 python3 train.py --num-envs 1048576 --episodes 1000 2>&1 | grep -v amdgpu.ids | grep -v "^epoch [0-9]*[1-9]/" | grep -v "^epoch [0-9]*[1234678]0/" | cut -c1-260 | tee $OUT/train_1048576x1000.log
echo "wall seconds: $((SECONDS - S))" | tee -a $OUT/train_1048576x1000.log
timeout -k 10 600 python3 train.py --num-envs 262144 --episodes 60 --board-size 5 2>&1 | grep -v amdgpu.ids | tail -n 3 | cut -c1-260
exit 0
