#!/bin/bash
# L1 (TCP) tag-lookup / stall counters on a reduced workload, one pass, counters only
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p "$OUT"
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python bench.py --steps 1 --warmup 1 --iters 3 --width 1552 --height 1032 --no-cpu-baseline > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "$name rc=$?"; }
run tcp1 TCP_TAGRAM0_REQ TCP_TAGRAM1_REQ TCP_TAGRAM2_REQ TCP_TAGRAM3_REQ
run tcp2 TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_GATE_EN1
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
