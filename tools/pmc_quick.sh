# quick PMC look at a few kernels of the default bench (GPU box, repo root): bash tools/pmc_quick.sh <tag> <kernel regex> [bench args]
cd $GRAFT_REPO_ROOT
T=$1; RE=$2; shift 2
timeout 1200 bash tools/pmc_collect.sh gpurun_out/pmcq_$T "$@" > /dev/null 2>&1
python tools/pmc_table.py gpurun_out/pmcq_$T gpurun_out/pmcq_$T.json > /dev/null
rm -rf gpurun_out/pmcq_$T
python - <<PY
import json, re
t = json.load(open("gpurun_out/pmcq_$T.json"))
for k, v in sorted(t["kernels"].items()):
    if not re.search(r"$RE", k): continue
    wc = max(v.get("SQ_WAVE_CYCLES", 1), 1)
    print(k)
    print("   waves %d  VALU insts %.3g  vmem_rd %.3g  vmem_wr %.3g  lds %.3g  salu %.3g" % (v.get("SQ_WAVES", 0), v.get("SQ_INSTS_VALU", 0), v.get("SQ_INSTS_VMEM_RD", 0), v.get("SQ_INSTS_VMEM_WR", 0), v.get("SQ_INSTS_LDS", 0), v.get("SQ_INSTS_SALU", 0)))
    print("   busy_cycles %.3g  wave_cycles %.3g  active_valu/wc %.2f  wait_any/wc %.2f  wait_inst_any/wc %.2f  lane_util %s" % (v.get("SQ_BUSY_CYCLES", 0), wc, v.get("SQ_ACTIVE_INST_VALU", 0) / wc, v.get("SQ_WAIT_ANY", 0) / wc, v.get("SQ_WAIT_INST_ANY", 0) / wc, v.get("lane_utilisation")))
    print("   TCC hit %.3g miss %.3g  fetch GB %.2f  write GB %.2f" % (v.get("TCC_HIT", 0), v.get("TCC_MISS", 0), v.get("fetch_bytes_per_launch_corrected_x2", 0) / 2e9, v.get("write_bytes_per_launch", 0) / 1e9))
PY
