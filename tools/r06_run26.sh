run() { DVP_MVS_LIB=$2 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-per-iteration 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], {k: round(v,1) for k,v in d['stage_ms_per_step'].items() if v > 20})"; }
run tree ""
for v in "$@"; do run $v $PWD/build/variants/$v.so; done
run tree ""
