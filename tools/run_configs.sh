#!/bin/bash
# Runs on the GPU box (through gpurun): the bench line of every BASELINE configuration.
# Output: gpurun_out/cfg/<config>.json (+ .err with the progress lines).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/cfg
mkdir -p "$O"
cd "$R"
set -e
for c in ${CONFIGS:-c2 c1 c3 c4 c5}; do
  extra=""
  [ "$c" = c4 ] && extra="--emulate-tp 8"
  echo "== $c $(date +%T)"
  timeout -k 10 ${CFG_TIMEOUT:-500} python3 bench.py --config $c $extra > "$O/$c.json" 2> "$O/$c.err"
  tail -c 1500 "$O/$c.json"
done
