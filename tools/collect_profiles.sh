#!/bin/bash
# Runs on the GPU box (through gpurun): bench line, rocprofv3 kernel statistics of the same
# command, and the two HBM-traffic counter passes of the decode kernel inside the bench.
# Output goes to gpurun_out/prof/; tools/summarize_profiles.py turns it into profiles/.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof
mkdir -p "$O"
cd /tmp
export TMPDIR=/tmp
timeout -k 10 400 python3 "$R/bench.py" --steps 3 --warmup 1 > "$O/bench_n1.json" 2> "$O/bench_n1.err"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -o bench -- \
    python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --alt-evict 0 > "$O/bench_under_rocprof.json" 2> "$O/stats.err"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -o fetch -- \
    python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --alt-evict 0 > /dev/null 2> "$O/pmc_fetch.err"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -o write -- \
    python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --alt-evict 0 > /dev/null 2> "$O/pmc_write.err"
ls -R "$O" | head -40
