mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs.py tests/test_random_configs.py -x -q -m gpu > gpurun_out/r06/parity_tests.log 2>&1
tail -2 gpurun_out/r06/parity_tests.log
timeout 900 bash tools/profile_bench.sh gpurun_out/r06 yzl --no-cpu-baseline > /dev/null
rm -rf gpurun_out/r06/trace_yzl
python -c "
import json; d=json.load(open('gpurun_out/r06/yzl_bench.json')); print(d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
grep "strong_decide\|strong_eval_items\|strong_refine" gpurun_out/r06/yzl_kernel_stats.txt | head -4
