#!/usr/bin/env bash
# A/B of library builds on the 4-call loop (tools/archive/exp_unfused.py), alternating processes on one box.
# usage: r03_ab_four_call.sh <name.so under tools/variants/ | product> ...
cd "$GRAFT_REPO_ROOT"
for r in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = product ]; then unset Q2048_LIB_PATH; else export Q2048_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/$v; fi
    timeout -k 10 200 python3 tools/archive/exp_unfused.py 2>/dev/null | python3 -c "
import sys, json
rows = [json.loads(l) for l in sys.stdin if l.startswith('{')]
print('$v', ' | '.join('%s B=%d %.1f us' % (r['api'][:6], r['B'], r['us_per_step']) for r in rows))"
  done
done
