# A/B of the weak update's forms on the GPU box (repo root): GPU parity tests of the weak path, then bench lines.
# usage: bash tools/ab_weak.sh <tag> [notest]
cd $GRAFT_REPO_ROOT
T=${1:-ab}
if [ "$2" != "notest" ]; then timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_edge_cases.py tests/test_random_configs.py -m gpu -q -x 2>&1 | tail -4; fi
run() { env $3 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $2 > gpurun_out/${T}_$1.json 2> gpurun_out/${T}_$1.err;
  python - <<PY
import json
d = json.load(open("gpurun_out/${T}_$1.json"))
print("$1", d["value"], d["ms_per_step"], d["config"]["weak_fraction"], {k: round(v, 1) for k, v in d["stage_ms_per_step"].items() if v > 5})
PY
}
run phased07 "" "X=1"
run g1111 "" "DVP_WEAK_GROUPS=1,1,1,1"
run g2488 "" "DVP_WEAK_GROUPS=2,4,8,8"
run g2888 "" "DVP_WEAK_GROUPS=2,8,8,8"
run g1244 "" "DVP_WEAK_GROUPS=1,2,4,4"
run phased25 "--weak-frac 0.25" "X=1"
bash tools/profile_bench.sh gpurun_out ${T}_trace --no-cpu-baseline > /dev/null 2>&1; rm -rf gpurun_out/trace_${T}_trace
grep -E "dvp_weak|^kernel" gpurun_out/${T}_trace_kernel_stats.txt
