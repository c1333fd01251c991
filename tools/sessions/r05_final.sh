#!/usr/bin/env bash
# round 5, closing session: smoke(), the whole GPU suite, the two bench lines again (the CPU twin changed: its
# figure under cpu_baseline.product_core), on the final sources
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05z; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
rc=$?; tail -n 6 $OUT/pytest_gpu.txt | cut -c1-300; echo "pytest rc $rc"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err; echo "bench k20 rc $?"
timeout -k 10 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python3 - <<'PY'
import json
for f in ("bench_k20.json", "bench.json"):
    d = json.loads([l for l in open("gpurun_out/r05z/" + f) if l.startswith("{")][-1]); r = d["roofline"]; cb = d["cpu_baseline"]
    print(f, "value %.4g us/step %.2f frac %.4f fabric_frac %s traffic %s" % (d["value"], d["ms_per_step"] * 1e3, r["frac"], r.get("fabric_frac"), r.get("traffic_bytes_per_env_step")))
    for c in d.get("companions", []): print("   ", c["name"], "%.4g %.1fus frac %.3f load %.3f" % (c["value"], c["ms_per_step"] * 1e3, c["roofline_frac"], c["table_load_factor"]))
    print("    cpu port %.3g @%d; product-core %.3g @%d, by threads %s" % (cb["value"], cb["cores"], cb["product_core"]["value"], cb["product_core"]["cores"], cb["product_core"]["by_threads"]))
PY
