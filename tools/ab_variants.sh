# A/B helper (GPU box, repo root): GPU tests, end-to-end timing, bench lines of the tree and of variant libraries
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
timeout 500 bash tools/e2e_timing.sh gpurun_out > gpurun_out/e2e.out 2>&1; grep -E "^pass|real" gpurun_out/e2e.out
run() { DVP_MVS_LIB=$2 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], {k: round(v,1) for k,v in d['stage_ms_per_step'].items() if v > 20})"; }
run tree_rotated "" ""
run tree_axis "" "--rig axis"
for v in "$@"; do run $v $PWD/build/variants/$v.so ""; done
