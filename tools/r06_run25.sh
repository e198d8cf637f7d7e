mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs.py -x -q -m gpu > gpurun_out/r06/parity_tests.log 2>&1
tail -2 gpurun_out/r06/parity_tests.log
timeout 900 bash tools/profile_bench.sh gpurun_out/r06 layout --no-cpu-baseline > /dev/null
rm -rf gpurun_out/r06/trace_layout
python -c "
import json; d=json.load(open('gpurun_out/r06/layout_bench.json')); print(d['value'], d['stage_ms_per_step'])"
grep "anchor_table\|ransac\|sweep_decide\|weak_eval" gpurun_out/r06/layout_kernel_stats.txt | head -12
