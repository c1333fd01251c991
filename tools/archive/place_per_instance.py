#!/usr/bin/env python3
"""Per-instance (XCD x L2 channel) view of rocprofv3 counters over tools/exp_place's 32 GiB buffers:
is a slow placement slow on SOME L2 channels / XCDs or on all of them alike?
Input: the `--output-format json` results of
    rocprofv3 --kernel-trace --pmc <base counters, not the _sum forms> -- tools/variants/exp_place a0:32 s0 ... a6:32 s6
(base counters keep their DIMENSION_XCC x DIMENSION_INSTANCE values; the CSV output sums them).
Usage: place_per_instance.py results.json[.gz] ..."""
import gzip
import json
import sys

import numpy as np

for path in sys.argv[1:]:
    raw = (gzip.open if path.endswith(".gz") else open)(path).read()
    d = json.loads(raw)["rocprofiler-sdk-tool"][0]
    names = {c["id"]["handle"]: c["name"] for c in d["counters"]}
    order = {c["id"]["handle"]: [(i["dimensions"][1]["index"], i["dimensions"][0]["index"]) for i in c["instances"]]
             for c in d["counters"]}
    ksym = {k["kernel_id"]: k.get("formatted_kernel_name") or k.get("kernel_name") for k in d["kernel_symbols"]}
    recs = [c for c in d["callback_records"]["counter_collection"]
            if "k_requests" in str(ksym.get(c["dispatch_data"]["dispatch_info"]["kernel_id"]))]
    print(f"== {path}: {len(recs)} k_requests dispatches = {len(recs) // 4} buffers x (1 warm-up + 3 timed)")
    for b in range(len(recs) // 4):
        grp = recs[b * 4 + 1:b * 4 + 4]
        dur = np.mean([c["dispatch_data"]["end_timestamp"] - c["dispatch_data"]["start_timestamp"] for c in grp]) / 64 / 1e3
        line = f"  buffer {b}: {dur:6.2f} us per 2^20 scattered stores"
        for h, nm in names.items():
            arr = np.zeros((8, 16))
            for c in grp:
                vals = [r["value"] for r in c["records"] if r["counter_id"]["handle"] == h]
                for (x, i), v in zip(order[h], vals):
                    arr[x, i] += v / 3
            perx, perc = arr.sum(1), arr.sum(0)
            line += (f" | {nm}: total {arr.sum():.3e}, max/mean over XCDs {perx.max() / perx.mean():.3f}, over L2 channels "
                     f"{perc.max() / perc.mean():.3f}, over the 128 instances {arr.max() / arr.mean():.3f} (min/mean {arr.min() / arr.mean():.3f})")
        print(line)
