#!/usr/bin/env bash
# A/B of builds of the library on one box: alternating short bench runs (main line only).
# usage: r03_ab_lib.sh "<bench args>" <name.so under tools/variants/ | product> ...
cd "$GRAFT_REPO_ROOT"
ARGS=$1; shift
for r in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = product ]; then unset Q2048_LIB_PATH; else export Q2048_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/$v; fi
    timeout -k 10 200 python3 bench.py --cpu-seconds 0 --no-companions --repeats 3 $ARGS 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value']/1e10, 4), round(d['roofline']['frac'], 4), round(d['ms_per_step']*1e3, 2))"
  done
done
