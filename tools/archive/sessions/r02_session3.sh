#!/usr/bin/env bash
# Round-2 GPU session 3: cache-policy flavours of the probe load (request size at the fabric).
set -u
TAG=${1:-r02c}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 300 tools/variants/exp_requests 32 20 32 32 "load" > "$OUT/flavours_cap32.json" 2> "$OUT/flavours.err"; rc=$?
cat "$OUT/flavours_cap32.json"; [ $rc -eq 124 ] && exit 1
echo "== request sizes per flavour (PMC)"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d "$OUT/pmc_rd" -- tools/variants/exp_requests 32 20 32 32 "load[" > "$OUT/pmc_rd.json" 2> "$OUT/pmc_rd.err"; rc=$?
echo "rc=$rc"; [ $rc -eq 124 ] && exit 1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
names = [r["requests"] for r in json.load(open(out + "/pmc_rd.json"))["rows"]]
acc = collections.defaultdict(dict)
for f in glob.glob(out + "/pmc_rd/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_requests" in r["Kernel_Name"]:
            acc[int(r["Dispatch_Id"])][r["Counter_Name"]] = acc[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(acc)
per = 4   # dispatches per combo: 1 warm-up + 3 timed
for k, name in enumerate(names):
    d = acc[ids[k * per + 1]]
    lanes_steps = (1 << 20) * 32
    print(name, {c.replace("TCC_EA0_", ""): round(v / lanes_steps, 3) for c, v in sorted(d.items())})
PY
exit 0
