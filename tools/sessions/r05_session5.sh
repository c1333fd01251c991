#!/usr/bin/env bash
# round 5, session 5: the whole GPU suite, then train.py with the first growth's table mapped before the clock
# starts, then the driver's bench command (new: steady-state companion, product-core CPU baseline)
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05e; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
rc=$?; tail -n 15 $OUT/pytest_gpu.txt | cut -c1-300; echo "pytest rc $rc"; [ $rc -eq 0 ] || exit $rc
T="timeout -k 10 600 python3 train.py"
filt() { grep -v "^epoch [0-9]*[1-9]/" | grep -v amdgpu.ids | tail -n 24; }
echo "== 262144 x 100, default (growing table: 2^30 first, the next one prefetched)"
$T --num-envs 262144 --episodes 100 --log $OUT/train_262144x100_growing.csv 2>&1 | filt | tee $OUT/train_262144x100_growing.log
echo "== 1048576 x 20, default"
$T --num-envs 1048576 --episodes 20 --log $OUT/train_1048576x20_growing.csv 2>&1 | filt | tee $OUT/train_1048576x20_growing.log
echo "== 262144 x 100, first capacity 2^28"
$T --num-envs 262144 --episodes 100 --initial-capacity-log2 28 --log $OUT/train_262144x100_growing_from28.csv 2>&1 | filt | tee $OUT/train_262144x100_growing_from28.log
echo "== 262144 x 100, fixed 2^32"
$T --num-envs 262144 --episodes 100 --capacity-log2 32 --log $OUT/train_262144x100_fixed32.csv 2>&1 | filt | tee $OUT/train_262144x100_fixed32.log
echo "== bench.py, the driver's command"
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err; echo "bench rc $?"
python3 - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r05e/bench_k20.json") if l.startswith("{")][-1])
print("value %.4g ms/step %.4f frac %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
for c in d.get("companions", []): print("  ", c["name"], "%.4g" % c["value"], "%.4f us" % (c["ms_per_step"]*1e3), "frac %.4f" % c["roofline_frac"], "load %.3f" % c["table_load_factor"])
cb=d["cpu_baseline"]; print("cpu port %.3g on %d; product-core %.3g on %d (1 thread %.3g)" % (cb["value"], cb["cores"], cb["product_core"]["value"], cb["product_core"]["cores"], cb["product_core"]["single_thread"]["value"]))
PY
