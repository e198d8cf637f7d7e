# instruction-cache counters of the bench kernels (GPU box, repo root)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o -E "\b(SQC_ICACHE[A-Z_]*|SQ_IFETCH[A-Z_]*|SQC_INST[A-Z_]*|SQ_INST_LEVEL[A-Z_]*|SQ_WAIT_INST_LDS|SQ_WAIT_IFETCH[A-Z_]*|SQ_INSTS_BRANCH|SQ_VALU_MFMA_BUSY_CYCLES|SQ_INST_CYCLES_VMEM[A-Z_]*|SQ_LDS_BANK_CONFLICT|SQ_LDS_IDX_ACTIVE|SQ_ACTIVE_INST_SCA|SQ_ACTIVE_INST_MISC|SQ_WAVE_DEP_WAIT|SQ_WAIT_INST_LDS)\b" | sort -u | tr '\n' ' ' > gpurun_out/avail_counters.txt
cat gpurun_out/avail_counters.txt
OUT=gpurun_out/pmc_ic
rm -rf $OUT; mkdir -p $OUT
run() { name=$1; shift; timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; }
run ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_ic/ic/**/*counter_collection.csv", recursive=True)
rows = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0]
    rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:8]:
    print(k, {c: "%.3g" % x for c, x in v.items()})
PY
rm -rf $OUT
