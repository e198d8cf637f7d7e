set -x
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs.py -x -q -m gpu -k "strong or cfg1 or cfg3_shaped or nine" > gpurun_out/r06/strong_tests.log 2>&1
tail -3 gpurun_out/r06/strong_tests.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-per-iteration > gpurun_out/r06/ab_pairs.json 2> gpurun_out/r06/ab_pairs.err
python -c "
import json; d=json.load(open('gpurun_out/r06/ab_pairs.json')); print('pairs', d['value'], d['stage_ms_per_step'])"
rocprofv3 --kernel-trace --stats -d gpurun_out/r06/prof_strong -o strong -- python bench.py --steps 1 --warmup 0 --no-secondary --no-cpu-baseline --no-per-iteration > /dev/null 2>&1
python tools/rocpd_summary.py gpurun_out/r06/prof_strong/strong_results.db | grep -i "strong\|calls"
