#!/usr/bin/env bash
# round 5: the claim-first probe on 5x5 (experiment bit 15): correctness against the oracle, then the A/B on one
# pre-filled table, alternating launches.  Hypothesis: >= 6 % at load < 0.1 (0.8 loads per step saved of ~4.2 requests).
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05x; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 python3 tools/exp_claim_first_check.py 2>&1 | tail -n 3 | tee $OUT/claim_first_check.txt
[ "${PIPESTATUS[0]}" -eq 0 ] || exit 1
timeout -k 10 600 python3 tools/exp_load_curve_prefilled.py 30 5 0,0x8000 > $OUT/claim_first_ab.jsonl 2> $OUT/err.txt; echo "rc $?"
python3 - <<'PY'
import json
for l in open("gpurun_out/r05x/claim_first_ab.jsonl"):
    d = json.loads(l); b = d["by_experiment_bits"]
    print("load %.2f  shipped %s  claim-first %s" % ((d["load_before"] + d["load_after"]) / 2, b.get("0x0"), b.get("0x8000")))
PY
