#!/bin/bash
# Builds the CPU checker (oracle/pe_oracle.c) with the undefined-behaviour sanitizer (traps on
# signed overflow, misaligned or out-of-range accesses the compiler can see, bad shifts, float
# casts out of range) and runs the CPU test suite through that copy.  Host code only: no GPU.
set -euo pipefail
cd "$(dirname "$0")/.."
out=oracle/_san
mkdir -p "$out"
gcc -O1 -g -fPIC -shared -std=c11 -fno-fast-math -ffp-contract=off \
    -fsanitize=undefined,bounds-strict,float-cast-overflow -fno-sanitize-recover=all \
    -o "$out/libpe_oracle.so" oracle/pe_oracle.c -lm
PE_ORACLE_LIB="$PWD/$out/libpe_oracle.so" UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    python -m pytest tests -x -q -m "not gpu" "$@"
