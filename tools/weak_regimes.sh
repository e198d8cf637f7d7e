# The weak path at the WEAK shares the schedule really produces: bench lines at 6.9 % (default), 25 % and,
# on a 1552x1032 view, 90 % WEAK.  usage (GPU box, repo root): bash tools/weak_regimes.sh <tag>   -> gpurun_out/<tag>_weak*.json
cd $GRAFT_REPO_ROOT
T=${1:-r06}
run() { timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $2 > gpurun_out/${T}_$1.json 2> gpurun_out/${T}_$1.err;
  python - <<PY
import json
d = json.load(open("gpurun_out/${T}_$1.json"))
print("$1", d["value"], d["ms_per_step"], d["config"]["weak_fraction"], {k: round(v, 1) for k, v in d["stage_ms_per_step"].items() if v > 5})
PY
}
run weak07 ""
run weak25 "--weak-frac 0.25"
run weak90_1552 "--width 1552 --height 1032 --weak-frac 0.9"
run weak25_3104 "--width 3104 --height 2064 --weak-frac 0.25"
