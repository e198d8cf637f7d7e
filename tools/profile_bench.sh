#!/bin/bash
# rocprofv3 kernel trace of a bench run; summary via tools/rocpd_summary.py
# usage: tools/profile_bench.sh OUTDIR TAG [bench.py args]     (on the GPU box, from the repo root)
OUT=$1; TAG=$2; shift 2
export TMPDIR=/tmp
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats -d "$OUT/trace_$TAG" -o "$TAG" -- python bench.py --steps 2 --warmup 1 --no-secondary --no-per-iteration "$@" > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
echo "rc=$?"
DB=$(find "$OUT/trace_$TAG" -name "*.db" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-secondary --no-per-iteration $*   (MI355X)"; echo "# bench line of the same run: profiles/${TAG}_bench.json"; python tools/rocpd_summary.py "$DB"; } > "$OUT/${TAG}_kernel_stats.txt"
tail -1 "$OUT/${TAG}_bench.json"
