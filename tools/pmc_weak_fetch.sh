#!/bin/bash
OUT=$1; export TMPDIR=/tmp; mkdir -p "$OUT"
run() { name=$1; shift; timeout 500 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python tools/weak_pass_timing.py 3104 2064 5 2 0.10 > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "$name rc=$?"; }
run fetch FETCH_SIZE
run tcc TCC_HIT TCC_MISS TCC_REQ
