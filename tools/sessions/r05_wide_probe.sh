#!/usr/bin/env bash
# round 5: A/B of the two-slots-at-a-time probe (experiment bit 15, measurement build) against the shipped probe on one
# pre-filled table, alternating launches, loads 0.02 ... 0.71.  Hypothesis: >= 10 % at load 0.45, where 45 % of the
# probes need a second slot; loss at low load (one wasted request per probe).
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05w; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python3 tools/exp_load_curve_prefilled.py 30 4 0,0x8000 > $OUT/wide_probe_ab.jsonl 2> $OUT/err.txt; echo "rc $?"
python3 - <<'PY'
import json
for l in open("gpurun_out/r05w/wide_probe_ab.jsonl"):
    d = json.loads(l); b = d["by_experiment_bits"]
    print("load %.2f  shipped %s  wide %s" % ((d["load_before"] + d["load_after"]) / 2, b.get("0x0"), b.get("0x8000")))
PY
