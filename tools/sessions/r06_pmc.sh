#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"
bash tools/pmc_session.sh r06k20 --steps 20 --warmup 5 2>&1 | tail -n 30
bash tools/pmc_session.sh r06def --cap-log2 32 2>&1 | tail -n 30
