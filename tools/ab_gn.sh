# A/B of the GenNeighbours search kernel (GPU box, repo root): bench lines of the tree with the per-lane search and with the
# wave-per-pixel search (DVP_GN_WAVE=1), then of variant libraries under build/variants: tools/ab_gn.sh v1 v2 ...
cd $GRAFT_REPO_ROOT
[ -n "$GN_TESTS" ] && timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
run() { DVP_MVS_LIB=$2 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], {k: round(v,1) for k,v in d['stage_ms_per_step'].items() if v > 20})"; }
run tree ""
DVP_GN_WAVE=1 run wave ""
for v in "$@"; do run $v $PWD/build/variants/$v.so; done
