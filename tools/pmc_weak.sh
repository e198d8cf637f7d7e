#!/bin/bash
# SQ PMC passes over the weak-path timing script on a reduced view
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p "$OUT"
run() { name=$1; shift; timeout 400 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python tools/weak_pass_timing.py 1552 1032 5 2 0.10 > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "$name rc=$?"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
run sq2 SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
