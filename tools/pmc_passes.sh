#!/bin/bash
# PMC passes over one short bench run each (counters only: no --kernel-trace / --stats in the same
# run).  usage: tools/pmc_passes.sh <outdir> [bench args...]
set -u
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p "$OUT"
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python bench.py --steps 1 --warmup 1 --iters 2 --no-cpu-baseline $BENCH_ARGS > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "$name rc=$?"; }
BENCH_ARGS="$*"
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
run sq2 SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE TCC_HIT TCC_MISS
run tcp TCP_TOTAL_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES
run ta TA_BUSY TA_FLAT_READ_WAVEFRONTS TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES
find "$OUT" -name "*counter_collection.csv" | head
