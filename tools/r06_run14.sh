set -x
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs.py tests/test_random_configs.py -x -q -m gpu > gpurun_out/r06/tab_tests.log 2>&1
tail -3 gpurun_out/r06/tab_tests.log
timeout 600 python bench.py --steps 5 --warmup 1 --no-secondary --no-cpu-baseline --no-per-iteration > gpurun_out/r06/ab_sterm.json 2> gpurun_out/r06/ab_sterm.err
python -c "
import json; d=json.load(open('gpurun_out/r06/ab_sterm.json')); print('sterm', d['value'], d['stage_ms_per_step'])"
