#!/usr/bin/env bash
# Local pre-flight of a GPU session: rebuild every native library that is older than its sources (the .so files travel
# to the GPU box as they are), then hand the session script to gpurun.
#   tools/run_session.sh tools/sessions/r05_sessionN.sh [gpurun timeout, default 1200]
set -e
cd "$(dirname "$0")/.."
python3 -c "import __graft_entry__ as g; g.build()" | tail -n 1
exec gpurun --timeout "${2:-1200}" -- bash "$1"
