#!/usr/bin/env bash
# round 6: the 4-call loop of round 5's tree against this round's, alternating, on one box; then per-kernel times of both
# (tools/variants/r05tree: `git worktree add tools/variants/r05tree 6fb046c && make -C tools/variants/r05tree`; removed afterwards)
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r06f; mkdir -p $OUT
export TMPDIR=/tmp
for k in 1 2 3; do
  (cd tools/variants/r05tree && timeout -k 10 200 python3 tools/exp_unfused.py 2>/dev/null | head -n 2 | cut -c1-140 | sed 's/^/r05 /') | tee -a $OUT/ab.txt
  timeout -k 10 200 python3 tools/archive/exp_unfused.py 2>/dev/null | head -n 2 | cut -c1-140 | sed 's/^/r06 /' | tee -a $OUT/ab.txt
done
(cd tools/variants/r05tree && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d ../../../$OUT/prof_r05 -- python3 tools/exp_unfused.py > /dev/null 2>&1)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r06 -- python3 tools/archive/exp_unfused.py > /dev/null 2>&1
for t in r05 r06; do find $OUT/prof_$t -name "*kernel_stats.csv" | head -n 1 | while read -r f; do echo "== $t"; cut -d, -f1-4 "$f" | head -n 8 | cut -c1-160; cp "$f" $OUT/kernel_stats_$t.csv; done; rm -rf $OUT/prof_$t; done
