#!/usr/bin/env bash
# round 6, closing session A (final kernel sources): the GPU suite, the counter passes (the 4-call loop against round 5's tree ran in an earlier version
# of this session: profiles/r06_four_call_ab.txt, tools/sessions/r06_fourcall_ab.sh)
# (tools/variants/r05tree was a git worktree of round 5's last commit, 6fb046c, built with `make`: `git worktree add
# tools/variants/r05tree 6fb046c`; removed again after the session)
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r06y; mkdir -p $OUT; rm -rf gpurun_out/r06k20 gpurun_out/r06def gpurun_out/r06fz
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
rc=$?; tail -n 4 $OUT/pytest_gpu.txt | cut -c1-300; echo "pytest rc $rc"; [ $rc -eq 0 ] || exit $rc
bash tools/pmc_session.sh r06k20 --steps 20 --warmup 5 2>&1 | tail -n 3 | cut -c1-300
bash tools/pmc_session.sh r06def --cap-log2 32 2>&1 | tail -n 3 | cut -c1-300
bash tools/pmc_session.sh r06fz --steps 20 --warmup 5 --prefill-load 0.502 2>&1 | tail -n 3 | cut -c1-300
grep TRAFFIC_JSON gpurun_out/r06fz/pmc/summary.txt | sed 's/^TRAFFIC_JSON //' | python3 -c "import sys,json; print(json.dumps(json.loads(sys.stdin.read()), indent=1))" > $OUT/pmc_traffic_frozen_k20.json
grep -v TRAFFIC_JSON gpurun_out/r06fz/pmc/summary.txt | grep -v "k_table_probe\|k_table_export" > $OUT/pmc_summary_frozen_k20.txt
for t in k20 def; do
  n=$([ $t = k20 ] && echo _k20 || echo "")
  grep TRAFFIC_JSON gpurun_out/r06$t/pmc/summary.txt | sed 's/^TRAFFIC_JSON //' | python3 -c "import sys,json; print(json.dumps(json.loads(sys.stdin.read()), indent=1))" > $OUT/pmc_traffic$n.json
  grep -v TRAFFIC_JSON gpurun_out/r06$t/pmc/summary.txt | grep -v "k_table_probe\|k_table_export" > $OUT/pmc_summary$n.txt
done
exit 0
