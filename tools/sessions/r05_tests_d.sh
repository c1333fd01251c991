#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05v; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "no_room or falls_back or grows or train_default" > $OUT/pytest.txt 2>&1
rc=$?; tail -n 8 $OUT/pytest.txt | cut -c1-300; echo "pytest rc $rc"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-companions > $OUT/bench_k20_nocomp.json 2> $OUT/bench.err; echo "bench rc $?"
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05v/bench_k20_nocomp.json") if l.startswith("{")][-1]); cb = d["cpu_baseline"]
print("value %.4g frac %.4f fabric %.3f; cpu port %.3g@%d product-core %.3g@%d %s" % (d["value"], d["roofline"]["frac"], d["roofline"].get("fabric_frac") or 0, cb["value"], cb["cores"], cb["product_core"]["value"], cb["product_core"]["cores"], cb["product_core"]["by_threads"]))
PY
