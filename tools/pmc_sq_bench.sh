#!/bin/bash
# SQ PMC passes on the bench workload (3 iterations), counters only
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p "$OUT"
run() { name=$1; shift; timeout 500 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python bench.py --steps 1 --warmup 1 --iters 3 --no-cpu-baseline > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "$name rc=$?"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
run sq2 SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM
