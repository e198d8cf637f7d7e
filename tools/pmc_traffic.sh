#!/bin/bash
# HBM traffic of the bench workload's kernels: FETCH_SIZE and WRITE_SIZE in separate PMC passes
# (TCC has 4 slots: FETCH_SIZE takes 3, WRITE_SIZE 2), counters only, on the bench workload itself
# (6 iterations: per-launch averages over the same mix of early and converged iterations).
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p "$OUT"
run() { name=$1; shift; timeout 500 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "$name rc=$?"; }
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT TCC_MISS
