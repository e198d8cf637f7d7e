#!/bin/bash
# PMC passes over the bench workload (counters only, one rocprofv3 run per counter group: TCC has 4
# slots — FETCH_SIZE takes 3, WRITE_SIZE 2 — SQ has 8), then tools/pmc_table.py folds the per-launch
# averages of the TIMED step into profiles/pmc_r06.json keyed by kernel|WxH|S.
# usage (GPU box, repo root): tools/pmc_collect.sh OUTDIR [bench.py args, e.g. --config cfg3]
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p "$OUT"
run() { name=$1; ctrs=$2; shift 2; timeout 900 rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/$name" -o "$name" -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-per-iteration "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "$name rc=$?"; }
run fetch "FETCH_SIZE" "$@"
run write "WRITE_SIZE TCC_HIT TCC_MISS" "$@"
run sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "$@"
run sq2 "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" "$@"
# lane utilisation: active lanes x cycles of VALU work against the VALU-busy cycles of the same pass
run sq3 "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "$@"
# L1: tag look-ups of the vector L1s and their read requests to the L2 (round 6)
run tcp "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ" "$@"
