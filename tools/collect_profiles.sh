#!/bin/bash
# Runs on the GPU box (through gpurun): for the BASELINE configurations the bench line, the
# rocprofv3 kernel statistics of the same command and (c2 to c5) the two HBM-traffic counter passes
# of the decode kernel inside the bench; for the prefill kernel the SQ counter passes on the two
# shapes it is tracked on.  Output goes to gpurun_out/prof/; tools/summarize_profiles.py turns
# it into profiles/.   CONFIGS="c2 c3" selects configurations, PREFILL=0 skips the prefill passes.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof
rm -rf "$O"; mkdir -p "$O"
cd /tmp
export TMPDIR=/tmp
set -e
for c in ${CONFIGS:-c2 c1 c3 c4 c5}; do
  extra=""
  [ "$c" = c4 ] && extra="--emulate-tp 8"
  echo "== $c bench $(date +%T)"
  timeout -k 10 500 python3 "$R/bench.py" --config $c $extra > "$O/bench_$c.json" 2> "$O/bench_$c.err"
  short="--config $c $extra --no-cpu-baseline --alt-evict 0"
  if true; then
    echo "== $c kernel stats $(date +%T)"
    rshort="$short"
    [ "$c" = c5 ] && rshort="$short --output-len 256"    # the trace of 2048 steps x 32 layers is not needed
    timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_$c" -o bench -- \
        python3 "$R/bench.py" $rshort > "$O/bench_under_rocprof_$c.json" 2> "$O/stats_$c.err"
    rm -f "$O/stats_$c/bench_kernel_trace.csv"
  fi
  if [ "$c" = c2 ] || [ "$c" = c3 ] || [ "$c" = c4 ] || [ "$c" = c5 ]; then
    echo "== $c FETCH_SIZE $(date +%T)"
    timeout -s KILL 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch_$c" -o fetch -- \
        python3 "$R/bench.py" $short --steps 1 --warmup 0 > /dev/null 2> "$O/pmc_fetch_$c.err"
    python3 "$R/tools/pmc_summary.py" paged_decode_mfma "$O/pmc_fetch_$c/fetch_counter_collection.csv" > "$O/pmc_fetch_$c.txt"
    rm -f "$O/pmc_fetch_$c/fetch_counter_collection.csv"
    echo "== $c WRITE_SIZE $(date +%T)"
    timeout -s KILL 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write_$c" -o write -- \
        python3 "$R/bench.py" $short --steps 1 --warmup 0 > /dev/null 2> "$O/pmc_write_$c.err"
    python3 "$R/tools/pmc_summary.py" paged_decode_mfma "$O/pmc_write_$c/write_counter_collection.csv" > "$O/pmc_write_$c.txt"
    rm -f "$O/pmc_write_$c/write_counter_collection.csv"
  fi
done
if [ "${PREFILL:-1}" = 1 ]; then
  for shape in "8 1024" "4 8192"; do
    set -- $shape
    tag="b$1_l$2"
    echo "== prefill $tag $(date +%T)"
    timeout -k 10 200 python3 "$R/tools/microbench_prefill.py" --batch $1 --lq $2 --lkv $2 > "$O/prefill_$tag.txt" 2>&1
    timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
        --output-format csv -d "$O/pmc_pf_$tag" -o p -- python3 "$R/tools/microbench_prefill.py" --batch $1 --lq $2 --lkv $2 --iters 5 > /dev/null 2> "$O/pmc_pf_$tag.err"
    python3 "$R/tools/pmc_summary.py" prefill_mfma "$O/pmc_pf_$tag/p_counter_collection.csv" > "$O/pmc_pf_$tag.txt"
    timeout -s KILL 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS \
        --output-format csv -d "$O/pmc_pf2_$tag" -o p -- python3 "$R/tools/microbench_prefill.py" --batch $1 --lq $2 --lkv $2 --iters 5 > /dev/null 2> "$O/pmc_pf2_$tag.err"
    python3 "$R/tools/pmc_summary.py" prefill_mfma "$O/pmc_pf2_$tag/p_counter_collection.csv" >> "$O/pmc_pf_$tag.txt"
    rm -rf "$O/pmc_pf_$tag" "$O/pmc_pf2_$tag"
  done
fi
echo "== done $(date +%T)"
ls "$O"
