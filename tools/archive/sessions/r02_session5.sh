#!/usr/bin/env bash
# Round-2 GPU session 5: deterministic mode + pipelined env step: parity, speed, learning A/B.
set -u
TAG=${1:-r02e}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
bad() { [ "$1" -eq 124 ] || [ "$1" -eq 137 ]; }
echo "== pytest -m gpu"
timeout -k 10 1100 python -m pytest tests -m gpu -q -rA -s > "$OUT/pytest_gpu.log" 2>&1; rc=$?
grep -E "passed|failed|FAILED|Error" "$OUT/pytest_gpu.log" | tail -n 15; echo "pytest rc=$rc"; bad $rc && exit 1
[ $rc -eq 0 ] || { tail -n 60 "$OUT/pytest_gpu.log"; exit 1; }
echo "== deterministic mode speed"
timeout -k 10 300 python tools/archive/exp_det.py > "$OUT/det.jsonl" 2> "$OUT/det.err"; rc=$?
cat "$OUT/det.jsonl"; tail -n 3 "$OUT/det.err"; bad $rc && exit 1
echo "== env step kernel"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/prof_env" -- python3 tools/archive/exp_env_step.py > "$OUT/env_step.jsonl" 2> "$OUT/env_step.err"; rc=$?
cat "$OUT/env_step.jsonl"; tail -n 3 "$OUT/env_step.err"; bad $rc && exit 1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/prof_env/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_env_step" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")
            acc[(name, int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (name, grid), v in sorted(acc.items()):
    v = sorted(v)[len(v) // 10: -len(v) // 10 or None]
    print(f"{name:40s} grid={grid:9d} n={len(v):4d} mean_us={sum(v) / len(v) / 1e3:8.2f}")
PY
echo "== learning-quality A/B"
timeout -k 10 900 python tools/exp_learning_ab.py > "$OUT/learning_ab.jsonl" 2> "$OUT/learning_ab.err"; rc=$?
cat "$OUT/learning_ab.jsonl" | cut -c1-400; tail -n 3 "$OUT/learning_ab.err"
exit 0
