#!/usr/bin/env bash
# Round-2 GPU session 4: fine-grained / uncached table allocations (fabric request size).
set -u
TAG=${1:-r02d}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
for mode in 1 3; do
  echo "== alloc mode $mode"
  timeout -k 10 200 tools/variants/exp_requests 30 20 32 30 "load" $mode > "$OUT/alloc${mode}_cap30.json" 2> "$OUT/alloc${mode}.err"; rc=$?
  cat "$OUT/alloc${mode}_cap30.json"; tail -n 2 "$OUT/alloc${mode}.err"; [ $rc -eq 124 ] && exit 1
  timeout -k 10 200 tools/variants/exp_requests 30 20 32 30 "store" $mode > "$OUT/alloc${mode}_store_cap30.json" 2>> "$OUT/alloc${mode}.err"; rc=$?
  cat "$OUT/alloc${mode}_store_cap30.json"; [ $rc -eq 124 ] && exit 1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d "$OUT/pmc_rd$mode" -- tools/variants/exp_requests 30 20 32 30 "load[s" $mode > "$OUT/pmc_rd$mode.json" 2> "$OUT/pmc_rd$mode.err"; rc=$?
  [ $rc -eq 124 ] && exit 1
  python3 - "$OUT" $mode <<'PY'
import csv, glob, sys, collections, json
out, mode = sys.argv[1], sys.argv[2]
names = [r["requests"] for r in json.load(open(f"{out}/pmc_rd{mode}.json"))["rows"]]
acc = collections.defaultdict(dict)
for f in glob.glob(f"{out}/pmc_rd{mode}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_requests" in r["Kernel_Name"]:
            acc[int(r["Dispatch_Id"])][r["Counter_Name"]] = acc[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(acc)
for k, name in enumerate(names):
    d = acc[ids[k * 4 + 1]]
    print(name, {c.replace("TCC_EA0_", ""): round(v / ((1 << 20) * 32), 3) for c, v in sorted(d.items())})
PY
done
exit 0
