cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
timeout 500 bash tools/e2e_timing.sh gpurun_out > gpurun_out/e2e.out 2>&1; tail -12 gpurun_out/e2e.out
run() { DVP_MVS_LIB=$2 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], {k: round(v,1) for k,v in d['stage_ms_per_step'].items() if v > 20})"; }
run v1_inline_clampfree ""
run v0_noclampfree $PWD/build/variants/v0_noclampfree.so
run v2_noinline $PWD/build/variants/v2_noinline.so
run v3_lbweak3 $PWD/build/variants/v3_lbweak3.so
