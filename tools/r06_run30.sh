mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sweep or fused or bands or cfg3" > gpurun_out/r06/sweep_tests.log 2>&1
tail -3 gpurun_out/r06/sweep_tests.log
for g in 0 16; do DVP_SWEEP_BAND_GB=$g timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-per-iteration 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('band_gb=$g', d['value'], d['ms_per_step'], {k: round(v,1) for k,v in d['stage_ms_per_step'].items() if v > 20})"; done
E2E_CHECK_FUSION=1 timeout 1200 bash tools/e2e_timing.sh gpurun_out/r06 > gpurun_out/r06/e2e_console.log 2>&1
grep "^pass\|real\|identical" gpurun_out/r06/e2e_apd.txt
grep -n "Cost time" gpurun_out/r06/e2e_apd.log | awk -F'Cost time: ' '{print $2}' | awk '{printf "%s ", $1} END{print ""}'
