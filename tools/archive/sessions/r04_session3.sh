#!/usr/bin/env bash
# Round 4, session 3: the 5x5 rollout's request pattern on 8 / 32 / 128 GiB chunked
# tables (tools/archive/exp_requests.hip, claim rate 0.86), the 5x5 rollout itself on the same footprints, and the
# rollouts with Philox4x32-7 instead of -10 (tools/variants/libq2048_p7.so: -DQ2048_PHILOX_ROUNDS=7).
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04c; mkdir -p $OUT
export TMPDIR=/tmp
for cap in 28 30 32; do
  timeout -k 10 300 tools/variants/exp_requests $cap 20 64 $cap "cas+" 4 881 > $OUT/requests_5x5_chunks_cap$cap.json 2>$OUT/err.log || { tail -3 $OUT/err.log; exit 1; }
  python3 - <<PY
import json
d = json.load(open("$OUT/requests_5x5_chunks_cap$cap.json"))
print("cap 2^$cap chunks, claim rate 0.86:", {r["requests"]: r["us"] for r in d["rows"]})
PY
done
timeout -k 10 300 tools/variants/exp_requests 30 20 64 30 "cas+" 4 1024 > $OUT/requests_5x5_chunks_cap30_rate1.json 2>>$OUT/err.log
python3 -c "
import json; d = json.load(open('$OUT/requests_5x5_chunks_cap30_rate1.json')); print('cap 2^30 chunks, claim rate 1.0:', {r['requests']: r['us'] for r in d['rows']})"
for cap in 28 30 32; do
  for n in 5 4; do
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-companions --repeats 3 --board-size $n --cap-log2 $cap 2>>$OUT/err.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('rollout ${n}x${n} cap 2^$cap', 'us_per_step', round(d['ms_per_step']*1e3, 2), 'launch_us_per_step', round(r['avg_launch_ms']*1e3/20, 2), 'frac', round(r['frac'], 4), 'inserts/step', round(d['stats']['inserts_per_step'], 3), 'placement', d['config']['table_placement'].get('mode'), d['config']['table_placement'].get('probe_us'))" | tee -a $OUT/rollout_by_footprint.txt
  done
done
for n in 5 4; do
  bash tools/archive/sessions/r03_ab_lib.sh "--steps 20 --warmup 5 --board-size $n" product libq2048_p7.so 2>&1 | sed "s/^/philox ${n}x${n}: /" | tee -a $OUT/philox7_ab.txt
done
bash tools/archive/sessions/r03_ab_lib.sh "--steps 20 --warmup 5" product libq2048_w4.so libq2048_w5.so 2>&1 | sed "s/^/waves per SIMD (driver cmd): /" | tee -a $OUT/waves_ab.txt
timeout -k 10 400 python3 tools/archive/exp_intercept.py 2>>$OUT/err.log | tee $OUT/intercept.jsonl
INTERCEPT_BOARDS=65536 timeout -k 10 300 python3 tools/archive/exp_intercept.py 2>>$OUT/err.log | tee $OUT/intercept_65536.jsonl
FUZZ_PRODUCT=1 timeout -k 10 600 python3 tests/fuzz_parity.py 11 10 2>>$OUT/err.log | tail -3 | tee $OUT/fuzz_product_seed11.txt
