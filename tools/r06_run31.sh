mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs.py tests/test_fullsize_sampled_parity.py -x -q -m gpu > gpurun_out/r06/parity_tests.log 2>&1
tail -2 gpurun_out/r06/parity_tests.log
timeout 900 bash tools/profile_bench.sh gpurun_out/r06 cand --no-cpu-baseline > /dev/null
rm -rf gpurun_out/r06/trace_cand
python -c "
import json; d=json.load(open('gpurun_out/r06/cand_bench.json')); print(d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
grep "gen_candidates\|gen_neighbours\|gen_edge" gpurun_out/r06/cand_kernel_stats.txt | head -8
