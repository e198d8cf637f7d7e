# A/B of environment switches on the tree's library (GPU box, repo root): tools/ab_env.sh "NAME=VALUE ..." "NAME=VALUE ..." ...
# (an empty string = the defaults); optional variant libraries through DVP_MVS_LIB=build/variants/x.so in the same strings
cd $GRAFT_REPO_ROOT
run() { env $1 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-per-iteration 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$1]', d['value'], d['ms_per_step'], {k: round(v,1) for k,v in d['stage_ms_per_step'].items() if v > 15})"; }
for e in "$@"; do run "$e"; done
