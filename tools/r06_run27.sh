mkdir -p gpurun_out/r06
for e in 1 2; do DVP_SWEEP_SPLIT=$e timeout 400 python bench.py --config cfg2 --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --no-per-iteration 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('split=$e', d['value'], d['ms_per_step'], {k: round(v,1) for k,v in d['stage_ms_per_step'].items() if v > 5})"; done
DVP_SWEEP_SPLIT=2 timeout 1200 bash tools/e2e_timing.sh gpurun_out/r06 > gpurun_out/r06/e2e_console.log 2>&1
grep "^pass\|real" gpurun_out/r06/e2e_apd.txt
