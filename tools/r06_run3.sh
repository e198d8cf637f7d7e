# round 6, GPU call 3: the geometric term's fetch before the NCC evaluation (BASE) against round 5's order (GEOM_AFTER); valid store / table ablations; GPU suite
set -x
mkdir -p gpurun_out/r06
for v in BASE GEOM_AFTER NO_STORE TAB_CHEAP; do
  lib=$PWD/build/probe/abl_$v.so
  [ $v = BASE ] && lib=$PWD/dvp-mvs_amd/libdvp_mvs_hip.so
  DVP_MVS_LIB=$lib timeout 600 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-per-iteration > gpurun_out/r06/abl_$v.json 2> gpurun_out/r06/abl_$v.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r06/abl_$v.json"))
print("$v", d["value"], d["stage_ms_per_step"])
PY
done > gpurun_out/r06/sweep_ablation2.txt 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_fullsize_sampled_parity.py > gpurun_out/r06/gpu_suite.log 2>&1
tail -5 gpurun_out/r06/gpu_suite.log
