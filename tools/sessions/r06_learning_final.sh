#!/usr/bin/env bash
# round 6: the learning A/B on the final sources (the frozen modes run on line summaries)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06v
timeout -k 10 800 python3 tools/exp_learning_ab.py --modes store/64,cas,det,frozen,det-frozen 2>/dev/null > gpurun_out/r06v/learning_ab.jsonl
python3 - <<'PY'
import json
for l in open("gpurun_out/r06v/learning_ab.jsonl"):
    d = json.loads(l); print(d["mode"], d["seed"], round(d["last10_mean_return"], 1), round(d["last10_mean_score"]), d["table_rows"], d.get("drops"))
PY
