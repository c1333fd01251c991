#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"
bash tools/pmc_session.sh r05def --cap-log2 32 2>&1 | tail -n 12
