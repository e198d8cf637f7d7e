# round 6, GPU call 7: slot costs pixel-major + the decision launch staged through LDS, A/B against the rounds 3-5 layout
set -x
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs.py -x -q -m gpu -k "strong or cfg1 or cfg3_shaped or nine or golden" > gpurun_out/r06/strong_tests.log 2>&1
tail -4 gpurun_out/r06/strong_tests.log
run() { # name env...
  n=$1; shift
  env "$@" timeout 600 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-per-iteration > gpurun_out/r06/ab_$n.json 2> gpurun_out/r06/ab_$n.err
  python -c "
import json; d=json.load(open('gpurun_out/r06/ab_$n.json')); print('$n', d['value'], d['stage_ms_per_step']['strong_update'], d['stage_ms_per_step']['strong_prep'])"
}
run staged DVP_X=1
run pixelmajor_unstaged DVP_DECIDE_STAGED=0
run slotmajor DVP_MVS_LIB=$PWD/build/probe/slot0.so
rocprofv3 --kernel-trace --stats -d gpurun_out/r06/prof_strong -o strong -- python bench.py --steps 1 --warmup 0 --no-secondary --no-cpu-baseline --no-per-iteration > /dev/null 2>&1
python tools/rocpd_summary.py gpurun_out/r06/prof_strong 2>/dev/null | grep -i "strong\|name" | head -20
