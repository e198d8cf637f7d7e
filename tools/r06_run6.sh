# round 6, GPU call 6: candidates at anchor pixels only — full-size forms test, GPU suite, bench, ten-view schedule
set -x
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_fullsize_sampled_parity.py -q -m gpu -s -x > gpurun_out/r06/fullsize_parity.log 2>&1
tail -6 gpurun_out/r06/fullsize_parity.log
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_fullsize_sampled_parity.py > gpurun_out/r06/gpu_suite.log 2>&1
tail -4 gpurun_out/r06/gpu_suite.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-per-iteration > gpurun_out/r06/bench_mask.json 2> gpurun_out/r06/bench_mask.err
python -c "
import json; d=json.load(open('gpurun_out/r06/bench_mask.json')); print(d['value'], d['stage_ms_per_step'])"
timeout 1500 bash tools/e2e_timing.sh gpurun_out/r06 > gpurun_out/r06/e2e_console.log 2>&1
grep -n "^pass\|real\|fusion" gpurun_out/r06/e2e_apd.txt
