#!/bin/bash
# lane utilisation of the VALU work per kernel: SQ_THREAD_CYCLES_VALU (active lanes x cycles) against SQ_ACTIVE_INST_VALU x 64
# usage (GPU box, repo root): tools/pmc_lanes.sh OUTDIR [bench.py args]
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p "$OUT"
timeout 900 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d "$OUT/lanes" -o lanes -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > "$OUT/lanes.json" 2> "$OUT/lanes.err"
echo "lanes rc=$?"
python - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
cs = glob.glob(out + "/lanes/**/*counter_collection.csv", recursive=True)
rows = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
seen = set()
for r in csv.DictReader(open(cs[0])):
    k = r["Kernel_Name"].split("(")[0]
    rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (k, r["Dispatch_Id"]) not in seen:
        seen.add((k, r["Dispatch_Id"])); n[k] += 1
print("%-34s %5s %14s %10s %10s %10s" % ("kernel", "calls", "INSTS_VALU", "lane_util", "valu_busy", "wait_any"))
for k, v in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:16]:
    act = v.get("SQ_ACTIVE_INST_VALU", 0)
    lu = v.get("SQ_THREAD_CYCLES_VALU", 0) / (act * 64) if act else float("nan")
    wc = v.get("SQ_WAVE_CYCLES", 0)
    print("%-34s %5d %14.4g %10.3f %10.3f %10.3f" % (k[:34], n[k], v.get("SQ_INSTS_VALU", 0), lu, act / wc if wc else 0, v.get("SQ_WAIT_ANY", 0) / wc if wc else 0))
PY
