mkdir -p gpurun_out/r06
run() { DVP_MVS_LIB=$PWD/build/variants/$1.so timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-per-iteration 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], {k: round(v,1) for k,v in d['stage_ms_per_step'].items() if v > 20})"; }
for v in nt0 nt15 nt3 nt12 nt1 nt2 nt0; do run $v; done
